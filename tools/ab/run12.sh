L="tools/ab/bw.so tools/ab/am1.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'adam' $L
cp tools/ab/am1.so dqn_zoo_amd/libdqnzoo_hip.so
