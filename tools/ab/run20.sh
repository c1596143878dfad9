L="tools/ab/cur.so tools/ab/nt1.so tools/ab/nt2.so"
NB=2 bash tools/ab.sh libs 'fc_stream|adam|fc1_dgrad' $L
cp tools/ab/cur.so dqn_zoo_amd/libdqnzoo_hip.so
