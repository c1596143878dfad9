#!/bin/bash
# kernel trace of tools/act_check.py: durations of the decision kernel back to back and with a host wait
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/act; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/kt
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/tools/act_check.py > $OUT/kt.log 2>&1 < /dev/null
grep -v "^[EW]20" $OUT/kt.log | grep -v amdgpu.ids | cut -c1-140
python - <<P
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'act_one' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
s = [int(r['Start_Timestamp']) / 1e3 for r in rows]
print('n', len(d))
b2b = d[-980:-520]; hw = d[-480:]   # the script ends with 500 back-to-back and 500 awaited decisions
print('back to back: dur mean %.2f min %.2f max %.2f; start-to-start %.2f' % (sum(b2b)/len(b2b), min(b2b), max(b2b), (s[-521]-s[-980])/459))
print('host wait   : dur mean %.2f min %.2f max %.2f' % (sum(hw)/len(hw), min(hw), max(hw)))
P
