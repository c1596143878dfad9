mkdir -p gpurun_out/q
cp tools/ab/base.so dqn_zoo_amd/libdqnzoo_hip.so; python tools/ab/diag_qr3.py base 2>&1 | grep -E "CMP|no view|^conv"
for lib in c2_223 c3_223 c23; do
  cp tools/ab/$lib.so dqn_zoo_amd/libdqnzoo_hip.so
  echo "== $lib"
  python tools/ab/diag_qr3.py $lib base 2>&1 | grep -E "CMP|no view|^conv"
done
cp tools/ab/c23.so dqn_zoo_amd/libdqnzoo_hip.so
