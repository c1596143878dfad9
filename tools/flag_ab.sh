#!/bin/bash
# Same-box A/B of bench.py FLAG sets on the current library: bash tools/flag_ab.sh "<flags A>" "<flags B>" ...
# (each set: NB bench lines of 3000 steps; alternating order so that drift hits both)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BARGS="--cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0"
for i in $(seq 1 ${NB:-3}); do
  for f in "$@"; do
    timeout 300 python $R/bench.py --steps 3000 --warmup 300 $BARGS $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench [%s]' % sys.argv[1], d['value'], d['ms_per_step'])" "$f"
  done
done
