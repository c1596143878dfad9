#!/bin/bash
# kernel-trace only A/B of one env knob: bash tools/r3_ab2.sh VAR "v1 v2 .." pattern
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $2; do
export $1=$v
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 $BENCH_ARGS > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
echo "== $1=$v"
python $R/tools/step_trace_summary.py "$t" 200 | grep -E "busy|${3:-adam}" | cut -c1-110
rm -rf $OUT/kt
done
