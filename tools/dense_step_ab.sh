#!/bin/bash
# same-box A/B of library builds on the C51 / QR-DQN learner steps (tools/run_dense.py <kind> 12 prof)
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so; echo "== $lib"
  for w in ${KINDS:-c51 qr}; do timeout 100 python $R/tools/run_dense.py $w 12 prof 2>&1 | grep "us/step" | tail -2; done
done
cp /tmp/lib_keep.so $R/dqn_zoo_amd/libdqnzoo_hip.so
