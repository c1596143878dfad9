mkdir -p gpurun_out/q
bash tools/ab_check.sh tools/ab/base.so tools/ab/c1_21.so tools/ab/c1_41.so tools/ab/c1_42.so tools/ab/c1_81.so
NB=1 bash tools/ab.sh libs 'ConvFwdOp<1|conv1_dma' tools/ab/c1_21.so tools/ab/c1_41.so tools/ab/c1_42.so tools/ab/c1_81.so tools/ab/c1_22.so
cp tools/ab/c23.so dqn_zoo_amd/libdqnzoo_hip.so
python -m pytest tests -m gpu -x -q 2>&1 | tail -8
