#!/bin/bash
# One GPU session: the driver's bench command, a long sustained bench line, a rocprofv3
# kernel trace of the bench loop (default mode), two PMC passes (FETCH_SIZE, WRITE_SIZE;
# counters only, with --kernel-trace) on the learner step, the same two counters on
# tools/micro/stream_micro.bin (known byte counts: calibration of FETCH_SIZE for the
# dword-per-lane and float4-per-lane access patterns), and the SQ counter passes.
# Outputs under gpurun_out/<tag>/ (python tools/collect_profiles.py <tag> copies the
# summaries into profiles/).
ulimit -c 0
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err < /dev/null
echo "bench rc=$?"
timeout 600 python $R/bench.py --steps 40000 --warmup 500 --cpu-seconds 0 --other-configs 0 --prof-steps 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 > $OUT/bench_40k.json 2> $OUT/bench_40k.err < /dev/null
echo "bench 40k rc=$?"
for mode in fused sequential; do
  rm -rf $OUT/kt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $R/bench.py --mode $mode --steps 300 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 > $OUT/kt_$mode.log 2>&1 < /dev/null
  f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$mode.csv
  t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $R/tools/step_trace_summary.py "$t" 100 > $OUT/kernel_step_summary_$mode.txt 2>&1
  rm -rf $OUT/kt
done
# the one-launch-per-stage form of the head chain (dz_rainbow_args_t::separate_launches), same session
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --separate-launches --steps 300 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 > $OUT/kt_sep.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python $R/tools/step_trace_summary.py "$t" 100 > $OUT/kernel_step_summary_separate_launches.txt 2>&1
rm -rf $OUT/kt
pmc_table() {  # $1 = counter_collection.csv, $2 = counter name
python - "$1" "$2" <<'PY'
import csv, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
  if r.get('Counter_Name') == sys.argv[2]:
    d[r['Kernel_Name']].append(float(r['Counter_Value']))
print('kernel,%s_KB_median_per_launch,launches' % sys.argv[2])
for k, v in sorted(d.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
  v = sorted(v)
  print('"%s",%.1f,%d' % (k[:110], v[len(v) // 2], len(v)))
PY
}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -- python $R/tools/run_fwd.py > $OUT/pmc_$c.log 2>&1 < /dev/null
  f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && pmc_table "$f" $c > $OUT/pmc_$c.csv
  rm -rf $OUT/pmc
  if [ -x $R/tools/micro/stream_micro.bin ]; then
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -- $R/tools/micro/stream_micro.bin > $OUT/pmc_cal_$c.log 2>&1 < /dev/null
    f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && pmc_table "$f" $c > $OUT/pmc_cal_$c.csv
    rm -rf $OUT/pmc
  fi
done
# the dense learners (BASELINE configs 2 and 3): the same two counters
for cfg in dqn double_q; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -- python $R/tools/run_dense.py $cfg > $OUT/pmc_${cfg}_$c.log 2>&1 < /dev/null
    f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && pmc_table "$f" $c > $OUT/pmc_${cfg}_$c.csv
    rm -rf $OUT/pmc
  done
done
# the agent loop (act -> insert -> learn every 4th frame) and the dense learners' kernel trace
python $R/tools/agent_loop_bench.py 20000 rainbow json 2>/dev/null | tail -1 > $OUT/agent_loop_rainbow.json
python $R/tools/agent_loop_bench.py 20000 dqn json 2>/dev/null | tail -1 > $OUT/agent_loop_dqn.json
python $R/tools/agent_loop_bench.py 8000 iqn json 2>/dev/null | tail -1 > $OUT/agent_loop_iqn.json
# per-kernel durations of the DQN learner's loop (BASELINE config 2)
DZ_STEP_MARKER=finalize_grads bash $R/tools/trace_summary.sh 200 python $R/tools/run_dense.py dqn 400 > $OUT/kernel_step_summary_dqn.txt 2>&1
# in-kernel timeline of the multi-role head launch (a -DDZ_HC_STAMPS build of dz_rainbow.hip, if present)
if [ -f $R/tools/ab/hc_stamps.so ]; then
  cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/lib_keep.so; cp $R/tools/ab/hc_stamps.so $R/dqn_zoo_amd/libdqnzoo_hip.so
  timeout 200 python $R/tools/hc_stamps.py > $OUT/head_chain_stamps.txt 2>&1
  cp /tmp/lib_keep.so $R/dqn_zoo_amd/libdqnzoo_hip.so
fi
# the one-launch decision: parity print-out, back-to-back and per-decision latency, kernel durations
bash $R/tools/act_prof.sh > $OUT/act_decision.txt 2>&1
bash $R/tools/dense_trace.sh 200 > $OUT/kernel_step_summary_double_q.txt 2>&1
# the headline's step with the 18-action head: per-kernel averages of its loop
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/tools/run_rainbow_actions.py 18 400 > $OUT/kt_a18.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python $R/tools/step_trace_summary.py "$t" 200 > $OUT/kernel_step_summary_a18.txt 2>&1
rm -rf $OUT/kt
bash $R/tools/pmc_rainbow.sh > $OUT/pmc_sq_rainbow.txt 2>&1
# the IQN learner step: HIP-event duration of every launch, un-profiled step time, SQ counters
(timeout 150 python $R/tools/iqn_probe.py prof; timeout 150 python $R/tools/iqn_probe.py time) 2>&1 | grep -v amdgpu.ids > $OUT/iqn_step_launches.txt
bash $R/tools/pmc_iqn.sh > $OUT/pmc_sq_iqn.txt 2>&1
# the C51 / QR-DQN learner steps (HIP-event durations, steps per second)
for w in c51 qr; do timeout 150 python $R/tools/run_dense.py $w 12 prof 2>&1 | grep -v amdgpu.ids; done > $OUT/dense_c51_qr_steps.txt
ls -la $OUT | tail -20
head -3 $OUT/kernel_step_summary_fused.txt
