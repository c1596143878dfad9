L="tools/ab/f123.so tools/ab/b3_1123.so tools/ab/b3_2223.so tools/ab/b3_1143.so tools/ab/b3_1124.so tools/ab/b2_123.so tools/ab/b2_143.so tools/ab/b2_124.so"
bash tools/ab_check.sh $L
NB=1 bash tools/ab.sh libs 'Conv.*grad|ConvWg|ConvDg|dmaop' $L
cp tools/ab/f123.so dqn_zoo_amd/libdqnzoo_hip.so
