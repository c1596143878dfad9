L="tools/ab/pad3.so tools/ab/pad2.so"
NB=1 bash tools/ab.sh libs 'conv1_dma' $L
cp tools/ab/cur.so dqn_zoo_amd/libdqnzoo_hip.so
