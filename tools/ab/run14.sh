L="tools/ab/bw2.so tools/ab/c1_121.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'conv1_dma' $L
cp tools/ab/bw2.so dqn_zoo_amd/libdqnzoo_hip.so
