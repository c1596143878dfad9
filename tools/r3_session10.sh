#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
adam=[i for i,r in enumerate(rows) if 'adam' in r['Kernel_Name']]
print('adam launches', len(adam))
last=adam[-30:]
prev=None
for i in last:
  e=rows[i]['e']
  if prev is not None: print('step span %.1f us' % ((e-prev)/1e3))
  prev=e
# gaps in the last 21 steps
a=adam[-21]; seg=rows[a:]
big=[(seg[j+1]['s']-seg[j]['e'], seg[j]['Kernel_Name'][:40], seg[j+1]['Kernel_Name'][:40]) for j in range(len(seg)-1)]
big.sort(reverse=True)
print('largest gaps (us):', [(round(g/1e3,1),a,b) for g,a,b in big[:6]])
PY
tail -2 $OUT/kt.log | cut -c1-300
rm -rf $OUT/kt
