L="tools/ab/abl1.so tools/ab/abl2.so tools/ab/abl3.so tools/ab/abl4.so"
NB=0 bash tools/ab.sh libs 'Conv.*grad|ConvWg|ConvDg|dmaop' $L
cp tools/ab/f123.so dqn_zoo_amd/libdqnzoo_hip.so
