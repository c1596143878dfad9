#!/bin/bash
# An A/B build of the library with extra compiler flags:
#   bash tools/build_variant.sh <unit|all> <out.so> [-DFLAG=..]...
# <unit> = dz_rainbow / dz_dense / ... : only that translation unit is recompiled and linked with the
# other units' current objects (run `python -m dqn_zoo_amd.build` first); `all` recompiles everything.
unit=$1; out=$2; shift 2
R=/root/repo
CF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I $R/include"
tmp=$(mktemp -d)
if [ "$unit" = all ]; then
  for f in $R/dqn_zoo_amd/csrc/*.hip; do
    hipcc $CF "$@" -c $f -o $tmp/$(basename $f .hip).o 2>/dev/null &
  done
  wait
  hipcc --offload-arch=gfx950 -shared -fPIC -o $out $tmp/*.o && echo built $out
else
  hipcc $CF "$@" -c $R/dqn_zoo_amd/csrc/$unit.hip -o $tmp/$unit.o 2>$tmp/err || { echo "FAILED $out"; grep -m3 error $tmp/err; rm -rf $tmp; exit 1; }
  objs=$(ls $R/dqn_zoo_amd/csrc/_obj/*.o | grep -v "/$unit.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o $out $tmp/$unit.o $objs && echo built $out
fi
rm -rf $tmp
