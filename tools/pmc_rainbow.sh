ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pr_$tag -- python $R/tools/run_fwd.py > $OUT/pr_$tag.log 2>&1 < /dev/null
  f=$(find $OUT/pr_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
d = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
  d[r['Kernel_Name'][:125]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in sorted(d.items()):
  if 'at::' in k or 'rocclr' in k: continue
  print(k.replace('(anonymous namespace)::', ''), {n: int(sorted(v)[len(v) // 2]) for n, v in c.items()})
PY
  rm -rf $OUT/pr_$tag
done
