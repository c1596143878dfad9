#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s4
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sequential --no-graphs"
for v in "seq_eager:" "fused:--fused-sample" "seq_eager2:" "fused2:--fused-sample"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 300 $B $flags > $OUT/b_$name.json 2> $OUT/b_$name.err < /dev/null
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/b_$name.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])" 2>&1 | tail -1)"
done
for v in "seq:" "fused:--fused-sample"; do
  name=${v%%:*}; flags=${v#*:}
  rm -rf $OUT/kt_$name
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$name -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sequential --no-graphs $flags > $OUT/kt_$name.log 2>&1 < /dev/null
  t=$(find $OUT/kt_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/tools/step_trace_summary.py "$t" 200 > $OUT/trace_$name.txt 2>&1
  rm -rf $OUT/kt_$name
  cat $OUT/trace_$name.txt
done
