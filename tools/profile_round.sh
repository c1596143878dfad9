#!/bin/bash
# One GPU session: full bench line, rocprofv3 kernel trace of the bench step and
# two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only, with --kernel-trace) on
# the learner step.  Outputs under gpurun_out/ (copied into profiles/ by hand).
ulimit -c 0
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err < /dev/null
echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --steps 300 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --no-graphs > $OUT/kt.log 2>&1 < /dev/null
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_kernel_stats.csv
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python - "$t" > $OUT/${TAG}_kernel_step_summary.txt <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-101], idx[-1]
seg = rows[a + 1:b + 1]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
span = int(seg[-1]['End_Timestamp']) - int(rows[a]['End_Timestamp'])
print('last 100 steps, eager launches under rocprofv3 --kernel-trace')
print('per step: kernels %.1f  busy %.1f us  span %.1f us  gaps %.1f us' % (
    len(seg) / 100, busy / 1e5, span / 1e5, (span - busy) / 1e5))
d = defaultdict(lambda: [0, 0])
for r in seg:
  n = r['Kernel_Name'][:90]
  d[n][0] += 1; d[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1]):
  print('%6.2f us/step  x%.2f  %s' % (t / 1e5, c / 100, n))
PY
fi
rm -rf $OUT/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python $R/tools/run_fwd.py > $OUT/pmc_$c.log 2>&1 < /dev/null
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" $c > $OUT/${TAG}_pmc_$c.csv <<'PY'
import csv, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
  if r.get('Counter_Name') == sys.argv[2]:
    d[r['Kernel_Name']].append(float(r['Counter_Value']))
print('kernel,%s_KB_median_per_launch,launches' % sys.argv[2])
for k, v in sorted(d.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
  v = sorted(v)
  print('"%s",%.1f,%d' % (k[:110], v[len(v) // 2], len(v)))
PY
  fi
  rm -rf $OUT/pmc_$c
done
bash $R/tools/pmc_rainbow.sh > $OUT/${TAG}_pmc_sq_rainbow.txt 2>&1
ls -la $OUT | tail -12
head -3 $OUT/${TAG}_kernel_step_summary.txt
