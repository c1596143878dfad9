#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s12
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0"
for v in "seq:--mode sequential" "rp_ev:--mode two-stream --replay-only-prefetch" "rp_val:--mode two-stream --replay-only-prefetch --value-sync" "full_val:--mode two-stream --value-sync" "fused:"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 300 $B $flags > $OUT/b_$name.json 2> $OUT/b_$name.err < /dev/null
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/b_$name.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" 2>&1 | tail -1)"
  tail -2 $OUT/b_$name.err | cut -c1-200
done
