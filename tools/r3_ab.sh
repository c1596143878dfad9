#!/bin/bash
# A/B of one env knob: bash tools/r3_ab.sh VAR "v1 v2" [trace-pattern]
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in $2; do
export $1=$v
timeout 300 python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1=$v bench', d['value'], d['ms_per_step'])"
done; done
for v in $2; do
export $1=$v
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
echo "== $1=$v"
python $R/tools/step_trace_summary.py "$t" 200 | grep -E "last 200|busy|${3:-adam}" | cut -c1-130
rm -rf $OUT/kt
done
