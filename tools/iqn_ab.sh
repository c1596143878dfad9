#!/bin/bash
# same-box A/B of library builds on the IQN learner step (tools/iqn_probe.py time)
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so; echo "== $lib"
  timeout 100 python $R/tools/iqn_probe.py ${MODE:-time} 2>&1 | grep -E "${PAT:-learn us}"
done
cp /tmp/lib_keep.so $R/dqn_zoo_amd/libdqnzoo_hip.so
