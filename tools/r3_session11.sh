#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s11
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 500"
for p in 4 50 300 1000; do for i in 1 2 3; do
  timeout 300 $D --prime-steps $p > $OUT/b.json 2> $OUT/b.err < /dev/null
  echo "prime=$p $(python -c "import json,sys; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d['sustained']['value'])" 2>&1 | tail -1)"
done; done
