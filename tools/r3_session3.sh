#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0"
for v in "seq_eager:--sequential --no-graphs" "rp_eager:--replay-only-prefetch --no-graphs" "rp_graph:--replay-only-prefetch" "rp_eager_hostev:--replay-only-prefetch --no-graphs --host-scope-events"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 300 $B $flags > $OUT/b_$name.json 2> $OUT/b_$name.err < /dev/null
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/b_$name.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" 2>&1 | tail -1)"
done
D="python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sequential"
for v in "d_graph:" "d_eager:--no-graphs" "d_graph_p300:--prime-steps 300" "d_eager_p300:--no-graphs --prime-steps 300" "d_eager_p0:--no-graphs --prime-steps 0"; do
  name=${v%%:*}; flags=${v#*:}
  for i in 1 2 3; do
    timeout 300 $D $flags > $OUT/b_$name.json 2> $OUT/b_$name.err < /dev/null
    echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/b_$name.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" 2>&1 | tail -1)"
  done
done
