#!/bin/bash
# kernel-trace A/B of bench.py argument sets on one box: bash tools/r3_ab3.sh pattern "args1" "args2" ...
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pat=$1; shift
for a in "$@"; do
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 $a > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
echo "== $a"
python $R/tools/step_trace_summary.py "$t" 200 | grep -E "busy|$pat" | cut -c1-110
rm -rf $OUT/kt
done
