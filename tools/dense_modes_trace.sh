R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for mode in fused sequential; do
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --mode $mode --steps 20 --warmup 5 --cpu-seconds 0 --prof-steps 0 --other-configs 1 --sustain-steps 0 --agent-form-steps 0 > $OUT/kt.log 2>&1 < /dev/null
t=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
echo "=== $mode (double-q)"; DZ_STEP_MARKER=finalize_grads python $R/tools/step_trace_summary.py "$t" 100 | cut -c1-110 | head -14
done
