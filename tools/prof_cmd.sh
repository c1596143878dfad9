cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt3 -- python $R/bench.py --steps 300 --warmup 50 --cpu-seconds 0 --prof-steps 0 --no-graphs > $R/gpurun_out/kt3.log 2>&1 < /dev/null
tail -2 $R/gpurun_out/kt3.log
f=$(find $R/gpurun_out/kt3 -name "*kernel_stats.csv" | head -1)
echo "stats: $f"
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/kt3_kernel_stats.csv; fi
t=$(find $R/gpurun_out/kt3 -name "*kernel_trace.csv" | head -1)
if [ -n "$t" ]; then python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take the last 100 steps worth of kernels: find adam kernels
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-101], idx[-1]
seg = rows[a + 1:b + 1]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
span = int(seg[-1]['End_Timestamp']) - int(rows[a]['End_Timestamp'])
print('per step: kernels %.1f  busy %.1f us  span %.1f us  gap %.1f us' % (len(seg) / 100, busy / 1e5, span / 1e5, (span - busy) / 1e5))
from collections import defaultdict
d = defaultdict(lambda: [0, 0])
for r in seg:
  n = r['Kernel_Name'][:70]
  d[n][0] += 1; d[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1]):
  print('%6.2f us/step  x%.2f  %s' % (t / 1e5, c / 100, n))
PY
fi
rm -rf $R/gpurun_out/kt3
