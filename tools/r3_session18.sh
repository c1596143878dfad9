#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in 64 2 1 64 2 1; do
DZ_TUNE_SG_CHUNKS=$c timeout 300 python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunks $c', d['value'], d['ms_per_step'])"
done
cd $R; python -m pytest tests/test_pipeline_gpu.py tests/test_replay_gpu.py -q 2>&1 | tail -2
