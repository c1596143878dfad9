#!/bin/bash
# kernel trace of tools/act_check.py: durations of the decision kernel back to back and with a host wait
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/act; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/kt
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/tools/act_check.py > $OUT/kt.log 2>&1 < /dev/null
tail -4 $OUT/kt.log
python - <<P
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'act_one' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
s = [int(r['Start_Timestamp']) / 1e3 for r in rows]
print('n', len(d))
b2b = d[60:540]; hw = d[-480:]
print('back to back: dur mean %.2f min %.2f max %.2f; start-to-start %.2f' % (sum(b2b)/len(b2b), min(b2b), max(b2b), (s[540]-s[60])/480))
print('host wait   : dur mean %.2f min %.2f max %.2f' % (sum(hw)/len(hw), min(hw), max(hw)))
P
