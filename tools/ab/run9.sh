L="tools/ab/f123.so tools/ab/bw.so tools/ab/w1_23.so tools/ab/w1_24.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'Conv|conv|dmaop' tools/ab/bw.so tools/ab/w1_23.so tools/ab/w1_24.so
cp tools/ab/bw.so dqn_zoo_amd/libdqnzoo_hip.so
