#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "0 12" "3 12" "3 16" "3 20" "3 24" "3 8"; do
  set -- $cfg
  rm -rf $OUT/kt
  DZ_TUNE_E=$1 DZ_TUNE_S=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 > $OUT/kt.log 2>&1 < /dev/null
  t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
  echo "== E=$1 S=$2"; python $R/tools/step_trace_summary.py "$t" 200 | grep -i "adam\|FcWgradOp<2, 2, 1, 2, 5>\|reduce_parts" | cut -c1-100
  rm -rf $OUT/kt
  DZ_TUNE_E=$1 DZ_TUNE_S=$2 timeout 300 python $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
