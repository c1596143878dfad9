#!/bin/bash
# rocprofv3 kernel trace of "$@", reduced to per-kernel average durations of the
# last N launches of each kernel:  bash tools/trace_summary.sh 200 python tools/act_trace.py
N=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt_tmp
mkdir -p $OUT/kt_tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_tmp -- "$@" > $OUT/kt_tmp.log 2>&1 < /dev/null
t=$(find $OUT/kt_tmp -name "*kernel_trace.csv" | head -1)
python - "$t" $N <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = defaultdict(list)
for r in rows:
  d[r['Kernel_Name'][:100]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1][-n:])):
  if len(v) < n // 2: continue
  v = v[-n:]
  print('%7.2f us  x%-4d %s' % (sum(v) / len(v) / 1e3, len(v), k))
  tot += sum(v) / n / 1e3
print('sum per iteration (over %d): %.1f us' % (n, tot))
last = rows[-1]; i0 = max(0, len(rows) - 8 * n)
span = (int(rows[-1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
print('span of the last %d launches: %.1f us' % (len(rows) - i0, span))
PY
rm -rf $OUT/kt_tmp
