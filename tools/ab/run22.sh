L="tools/ab/cur.so tools/ab/sideabl.so"
NB=0 bash tools/ab.sh libs 'conv1_dma' $L
cp tools/ab/cur.so dqn_zoo_amd/libdqnzoo_hip.so
