L="tools/ab/bw2.so tools/ab/hf256.so tools/ab/hf512.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'head_chain' $L
cp tools/ab/bw2.so dqn_zoo_amd/libdqnzoo_hip.so
