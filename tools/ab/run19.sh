L="tools/ab/nosplit.so tools/ab/split2.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'Conv|dmaop|wgrad' $L
cp tools/ab/nosplit.so dqn_zoo_amd/libdqnzoo_hip.so
