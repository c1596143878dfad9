#!/bin/bash
# Round-3 GPU session 1: GPU tests, then the sequential step vs the two-stream
# pipelined loop (device-scope vs default events, graphs vs eager), and a kernel
# trace of the pipelined loop.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s1
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3000 --warmup 300 --cpu-seconds 0 --prof-steps 0 --other-configs 0"
for v in "seq:--sequential" "pipe:" "pipe_hostev:--host-scope-events" "pipe_eager:--no-graphs" "seq_eager:--sequential --no-graphs"; do
  name=${v%%:*}; flags=${v#*:}
  timeout 300 $B $flags > $OUT/b_$name.json 2> $OUT/b_$name.err < /dev/null
  echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('$OUT/b_$name.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
done
# driver command, pipelined default
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/b_driver.json 2> $OUT/b_driver.err < /dev/null
echo "driver rc=$? $(python -c "import json; d=json.load(open('$OUT/b_driver.json')); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1)"
for v in "pipe:" "seq:--sequential"; do
  name=${v%%:*}; flags=${v#*:}
  rm -rf $OUT/kt_$name
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_$name -- python $R/bench.py --steps 400 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 $flags > $OUT/kt_$name.log 2>&1 < /dev/null
  t=$(find $OUT/kt_$name -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/tools/step_trace_summary.py "$t" 200 > $OUT/trace_$name.txt 2>&1
  rm -rf $OUT/kt_$name
  head -12 $OUT/trace_$name.txt
done
