#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --other-configs 0 > $OUT/b.json 2> $OUT/b.err < /dev/null
  echo "default(prof on) $(python -c "import json,sys; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['sustained']['value'])" 2>&1 | tail -1)"
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --other-configs 0 --prof-steps 0 > $OUT/b.json 2> $OUT/b.err < /dev/null
  echo "prof off $(python -c "import json,sys; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['sustained']['value'])" 2>&1 | tail -1)"
  timeout 300 python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --other-configs 0 --prof-steps 0 --prime-steps 4 > $OUT/b.json 2> $OUT/b.err < /dev/null
  echo "prime4 $(python -c "import json,sys; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['sustained']['value'])" 2>&1 | tail -1)"
done
