#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s14
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['ms_per_step'])"
done
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python $R/tools/step_trace_summary.py "$t" 200 | cut -c1-130
rm -rf $OUT/kt
