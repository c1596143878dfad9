L="tools/ab/bw2.so tools/ab/pf1.so tools/ab/pf0.so tools/ab/c2_243.so tools/ab/c2_242.so tools/ab/c3_262.so tools/ab/a1.so tools/ab/a2.so tools/ab/a3.so"
bash tools/ab_check.sh $L
NB=1 bash tools/ab.sh libs 'conv1_dma|conv_dma_fwd' $L
cp tools/ab/bw2.so dqn_zoo_amd/libdqnzoo_hip.so
