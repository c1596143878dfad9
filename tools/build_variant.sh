#!/bin/bash
# bash tools/build_variant.sh <out.so> [-DFLAG=..]...   (an A/B build of the library with extra flags)
out=$1; shift
R=/root/repo
tmp=$(mktemp -d)
for f in $R/dqn_zoo_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I $R/include "$@" -c $f -o $tmp/$b.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $tmp/*.o && echo built $out
rm -rf $tmp
