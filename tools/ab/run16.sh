L="tools/ab/cur.so tools/ab/ni2_113.so tools/ab/ni2_114.so tools/ab/ni2_123.so"
bash tools/ab_check.sh $L
NB=1 bash tools/ab.sh libs 'conv_dma_fwd' $L
cp tools/ab/cur.so dqn_zoo_amd/libdqnzoo_hip.so
