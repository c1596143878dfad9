#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s15
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/pytest.log | cut -c1-220
cd /tmp && export TMPDIR=/tmp
for m in fused sequential; do
timeout 600 python $R/bench.py --mode $m --steps 2000 --warmup 200 --cpu-seconds 0 --prof-steps 20 --sustain-steps 0 2> $OUT/b_$m.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], {k:(v['value'], v['config']['launch'][:12]) for k,v in d['other_configs'].items()})
for k,v in d['other_configs'].items(): print(k, {a:b['us'] for a,b in v['roofline'].get('per_kernel',{}).items()} if 'per_kernel' in v['roofline'] else v['roofline'])"
tail -2 $OUT/b_$m.err | cut -c1-300
done
timeout 600 python $R/bench.py --mode sequential --no-graphs --steps 2000 --warmup 200 --cpu-seconds 0 --prof-steps 0 --sustain-steps 0 2> $OUT/b_seq_eager.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seq eager', d['value'], d['ms_per_step'], {k:(v['value'], v['config']['launch'][:12]) for k,v in d['other_configs'].items()})"
