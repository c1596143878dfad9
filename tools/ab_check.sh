#!/bin/bash
# Correctness of library variants against the first one: bash tools/ab_check.sh tools/ab/base.so tools/ab/x.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/q
cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/lib_keep_chk.so
first=""
for lib in "$@"; do
  cp $R/$lib $R/dqn_zoo_amd/libdqnzoo_hip.so
  n=$(basename $lib .so)
  timeout 120 python $R/tools/ab_dump.py dump /tmp/dump_$n.npz > /dev/null 2>&1 || echo "DUMP FAILED $n"
  if [ -z "$first" ]; then first=$n; else echo "== $n vs $first"; python $R/tools/ab_dump.py cmp /tmp/dump_$first.npz /tmp/dump_$n.npz | grep -E "WORST|act2_0|feat_0|h1_0|params_1|grad_0"; fi
done
cp /tmp/lib_keep_chk.so $R/dqn_zoo_amd/libdqnzoo_hip.so
