ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pi_$tag -- python $R/tools/run_iqn.py > $OUT/pi_$tag.log 2>&1 < /dev/null
  f=$(find $OUT/pi_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
d = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
  if 'IqnLin' in r['Kernel_Name'] or 'IqnWgrad' in r['Kernel_Name']:
    d[r['Kernel_Name'][:60] + '|grid' + r.get('Grid_Size', '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in d.items():
  print(k, {n: sorted(v)[len(v) // 2] for n, v in c.items()})
PY
  rm -rf $OUT/pi_$tag
done
