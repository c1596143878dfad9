#!/bin/bash
# per-step spans of the driver's 20-step window (and the priming / warm-up steps in front of it)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 > $OUT/kt.log 2>&1 < /dev/null
tail -1 $OUT/kt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms', d['ms_per_step'])"
python - <<P
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ad = [r for r in rows if 'adam_onfly' in r['Kernel_Name']]
ends = [int(r['End_Timestamp']) / 1e3 for r in ad]
starts = [int(r['Start_Timestamp']) / 1e3 for r in ad]
print('adam launches', len(ad))
sp = [ends[i] - ends[i - 1] for i in range(1, len(ends))]
print('step spans (end of Adam to end of Adam), last 34:', ' '.join('%.1f' % x for x in sp[-34:]))
print('adam durations, last 26:', ' '.join('%.1f' % (ends[i] - starts[i]) for i in range(len(ad) - 26, len(ad))))
P
