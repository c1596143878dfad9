mkdir -p gpurun_out/q
for lib in base c23; do
  cp tools/ab/$lib.so dqn_zoo_amd/libdqnzoo_hip.so
  echo "== $lib"
  python -m pytest tests/test_dense_gpu.py -q -x -k "other_head_widths" 2>&1 | tail -4
done
bash tools/ab_check.sh tools/ab/base.so tools/ab/c1_5.so tools/ab/c1_2.so tools/ab/c1_1.so tools/ab/c1_4.so
NB=1 bash tools/ab.sh libs 'ConvFwdOp<1|conv1_dma' tools/ab/base.so tools/ab/c1_5.so tools/ab/c1_2.so tools/ab/c1_1.so tools/ab/c1_4.so
cp tools/ab/c23.so dqn_zoo_amd/libdqnzoo_hip.so
