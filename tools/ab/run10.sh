L="tools/ab/bw.so tools/ab/fc_4.so tools/ab/fc_5.so tools/ab/fc_6.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'fc_stream' $L
cp tools/ab/bw.so dqn_zoo_amd/libdqnzoo_hip.so
