#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 500"
for v in "spin1:--spin-sync 1" "spin0:--spin-sync 0" "spin1b:--spin-sync 1" "spin0b:--spin-sync 0" "spin1_100:--spin-sync 1 --steps 100" "spin0_100:--spin-sync 0 --steps 100"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 300 $D $flags > $OUT/b_$name.json 2> $OUT/b_$name.err < /dev/null
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/b_$name.json')); print(d['steps'], d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['sustained']['value'])" 2>&1 | tail -1)"
done
