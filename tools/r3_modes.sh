#!/bin/bash
# the bench loop's modes back to back on one box (3000 steps each)
cd /tmp && export TMPDIR=/tmp
for a in "--mode fused" "--mode sequential" "--mode sequential --graphs" "--mode two-stream" "--mode two-stream --graphs" "--mode fused --stored-gradients" "--mode fused"; do
timeout 300 python /root/repo/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$a]', d['value'], d['ms_per_step'])"
done
