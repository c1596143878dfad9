#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s7
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/b_driver.json 2> $OUT/b_driver.err < /dev/null
echo "driver rc=$?"; tail -3 $OUT/b_driver.err
python - <<PY
import json
d=json.load(open('$OUT/b_driver.json'))
print(d['value'], d['ms_per_step'], d.get('sustained'), d['config']['mode'], d['config']['launch'])
print(json.dumps(d['roofline']['step']))
print({k:v['us'] for k,v in d['roofline']['per_kernel'].items()})
print({k:(v['value']) for k,v in d.get('other_configs',{}).items()})
print(d.get('cpu_baseline',{}).get('value'), d.get('speedup_vs_cpu_baseline'))
PY
