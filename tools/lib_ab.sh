#!/bin/bash
# Same-box A/B of two BUILDS of the library: bash tools/lib_ab.sh <pattern> tools/ab/base.so tools/ab/new.so ...
# (each is copied over dqn_zoo_amd/libdqnzoo_hip.so of the box's scratch copy in turn)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pat=$1; shift
BARGS="--cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 ${BENCH_ARGS:-}"
for lib in "$@"; do
  cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so
  echo "== $lib"
  for i in $(seq 1 ${NB:-2}); do
    timeout 300 python $R/bench.py --steps 3000 --warmup 300 $BARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
  done
  rm -rf $OUT/kt
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 $BARGS > $OUT/kt.log 2>&1 < /dev/null
  t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_trace_summary.py "$t" 200 | grep -E "last 200|busy|$pat" | cut -c1-120
  rm -rf $OUT/kt
done
