#!/bin/bash
# Same-box A/B on one gpurun box (box-to-box spread of one build is +-1.5 %: only pairs from ONE call
# are compared).  Every variant: NB (default 2) un-profiled bench lines of 3000 steps, then the
# per-kernel averages of a rocprofv3 kernel trace (last 200 steps) for kernels matching <pattern>.
#
#   bash tools/ab.sh libs  <pattern> tools/ab/base.so tools/ab/new.so ...   builds of the library
#   bash tools/ab.sh flags <pattern> "" "--separate-launches" ...           bench.py argument sets
#   bash tools/ab.sh env   <pattern> "VAR=0" "VAR=1" ...                    environment variants
#   bash tools/ab.sh dense tools/ab/base.so tools/ab/new.so ...             BASELINE configs 2 / 3
#   bash tools/ab.sh now   [pattern]                                        the current build alone
#   bash tools/ab.sh iqn   tools/ab/base.so tools/ab/new.so ...             the IQN learner step (tools/iqn_probe.py;
#                          MODE=time|prof, PAT=<grep pattern of the printed lines>)
#   bash tools/ab.sh steps tools/ab/base.so tools/ab/new.so ...             the C51 / QR-DQN learner steps
#                          (tools/run_dense.py <kind> 12 prof; KINDS="c51 qr")
# BENCH_ARGS adds bench.py arguments to every run; NB = bench lines per variant; TRACE=0 skips
# the kernel trace.  (A variant library is built with tools/build_variant.sh.)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
mode=$1; shift
BARGS="--cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 ${BENCH_ARGS:-}"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; }
one() {   # $1 = env assignments, $2 = extra bench flags, $3 = pattern
  for i in $(seq 1 ${NB:-2}); do
    env $1 timeout 300 python $R/bench.py --steps 3000 --warmup 300 $BARGS $2 2>/dev/null | line
  done
  [ "${TRACE:-1}" = 0 ] && return
  rm -rf $OUT/kt
  env $1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 $BARGS $2 > $OUT/kt.log 2>&1 < /dev/null
  t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_trace_summary.py "$t" 200 | grep -E "last 200|busy|$3" | cut -c1-150
  rm -rf $OUT/kt
}
case $mode in
  libs)  pat=$1; shift
         for lib in "$@"; do cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so; echo "== $lib"; one "" "" "$pat"; done ;;
  flags) pat=$1; shift
         for f in "$@"; do echo "== [$f]"; one "" "$f" "$pat"; done ;;
  env)   pat=$1; shift
         for v in "$@"; do echo "== $v"; one "$v" "" "$pat"; done ;;
  dense) for lib in "$@"; do
           cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so; echo "== $lib"
           for i in $(seq 1 ${NB:-2}); do
             timeout 300 python $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --prof-steps 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rainbow', d['value'], {k: v['value'] for k, v in d['other_configs'].items()})"
           done
         done ;;
  now)   one "" "" "${1:-.}" ;;
  iqn)   cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/lib_keep.so
         for lib in "$@"; do
           cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so; echo "== $lib"
           timeout 100 python $R/tools/iqn_probe.py ${MODE:-time} 2>&1 | grep -E "${PAT:-learn us}"
         done
         cp /tmp/lib_keep.so $R/dqn_zoo_amd/libdqnzoo_hip.so ;;
  steps) cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/lib_keep.so
         for lib in "$@"; do
           cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so; echo "== $lib"
           for w in ${KINDS:-c51 qr}; do timeout 100 python $R/tools/run_dense.py $w 12 prof 2>&1 | grep "us/step" | tail -2; done
         done
         cp /tmp/lib_keep.so $R/dqn_zoo_amd/libdqnzoo_hip.so ;;
  *)     echo "usage: see the header of tools/ab.sh"; exit 2 ;;
esac
