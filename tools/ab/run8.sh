L="tools/ab/o3.so tools/ab/o4.so tools/ab/o3b.so tools/ab/o4b.so tools/ab/o5b.so"
NB=1 bash tools/ab.sh libs 'Conv|conv|dmaop' $L
cp tools/ab/f123.so dqn_zoo_amd/libdqnzoo_hip.so
