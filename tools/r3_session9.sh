#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in 0 1 2 3; do
  rm -rf $OUT/kt
  DZ_TUNE_E=$e timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 > $OUT/kt.log 2>&1 < /dev/null
  t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
  echo "== DZ_TUNE_E=$e"; python $R/tools/step_trace_summary.py "$t" 200 | grep -i "last 200\|adam\|FcWgradOp<2, 2, 1, 2, 5>" | cut -c1-110
  rm -rf $OUT/kt
  DZ_TUNE_E=$e timeout 300 python $R/bench.py --steps 2000 --warmup 200 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
