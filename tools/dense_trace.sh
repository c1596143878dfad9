#!/bin/bash
# kernel trace of the dense (DQN / double-Q) bench loops: the last 200 steps of the trace
# are the last other_config (double-Q + prioritized); $1 = steps back to summarise
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 1 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
DZ_STEP_MARKER=finalize_grads python $R/tools/step_trace_summary.py "$t" ${1:-100} | cut -c1-150
rm -rf $OUT/kt
