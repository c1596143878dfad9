#!/bin/bash
# Same-box A/B of environment-variable variants of the CURRENT build:
#   bash tools/env_ab.sh <kernel-name pattern> "VAR=0" "VAR=1" ...
# per variant: two un-profiled bench lines (3000 steps) and the per-kernel averages of a
# rocprofv3 kernel trace (last 200 steps) for kernels matching the pattern.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pat=$1; shift
BARGS="--cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 ${BENCH_ARGS:-}"
for v in "$@"; do
  echo "== $v"
  for i in 1 2; do
    env $v timeout 300 python $R/bench.py --steps 3000 --warmup 300 $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
  done
  rm -rf $OUT/kt
  env $v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 $BARGS > $OUT/kt.log 2>&1 < /dev/null
  t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_trace_summary.py "$t" 200 | grep -E "last 200|busy|$pat" | cut -c1-120
  rm -rf $OUT/kt
done
