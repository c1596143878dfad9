#!/bin/bash
# Same-box A/B of library builds on BASELINE configs 2 and 3 (dense learners):
#   bash tools/dense_ab.sh tools/ab/base.so tools/ab/new.so ...
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so
  echo "== $lib"
  for i in 1 2; do
    timeout 300 python $R/bench.py --steps 1000 --warmup 100 --cpu-seconds 0 --prof-steps 0 --sustain-steps 0 --agent-form-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rainbow', d['value'], {k: v['value'] for k, v in d['other_configs'].items()})"
  done
done
