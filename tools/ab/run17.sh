L="tools/ab/cur.so tools/ab/sg1.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'adam|Conv.*grad.*1, 84|ConvWgradOp<1' $L
cp tools/ab/sg1.so dqn_zoo_amd/libdqnzoo_hip.so
python -m pytest tests/test_fused_step_gpu.py -q -x 2>&1 | tail -3
cp tools/ab/cur.so dqn_zoo_amd/libdqnzoo_hip.so
