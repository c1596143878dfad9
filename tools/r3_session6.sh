#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sequential --no-graphs --fused-sample"
for c in 64 4 2 1; do for ab in 1024 1536 1792 2048; do
  DZ_TUNE_SG_CHUNKS=$c DZ_TUNE_ADAM_BLOCKS=$ab timeout 300 $B > $OUT/b.json 2> $OUT/b.err < /dev/null
  echo "chunks=$c adam=$ab $(python -c "import json,sys; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
done; done
