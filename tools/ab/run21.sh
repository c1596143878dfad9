L="tools/ab/ph0.so tools/ab/ph3b.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'fc1_dgrad|dmaop|wgrad3' $L
cp tools/ab/ph3b.so dqn_zoo_amd/libdqnzoo_hip.so
python -m pytest tests/test_fused_step_gpu.py tests/test_head_chain_gpu.py tests/test_replay_gpu.py tests/test_agent_gpu.py tests/test_dense_agents_gpu.py -q -x 2>&1 | tail -3
