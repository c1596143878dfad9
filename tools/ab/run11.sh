L="tools/ab/bw.so tools/ab/s16_d3.so tools/ab/s16_d4.so tools/ab/s16_d6.so tools/ab/s16_dma4.so tools/ab/s16_dma8.so tools/ab/s32_d4.so"
bash tools/ab_check.sh $L
NB=2 bash tools/ab.sh libs 'fc_stream|head_chain' $L
cp tools/ab/bw.so dqn_zoo_amd/libdqnzoo_hip.so
