#!/bin/bash
# Builds several A/B variants of one translation unit in parallel:
#   bash tools/build_variants.sh dz_rainbow name1:"-DX=1 -DY=2" name2:"-DX=3" ...   -> tools/ab/<name>.so
unit=$1; shift
mkdir -p /root/repo/tools/ab
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  bash /root/repo/tools/build_variant.sh $unit /root/repo/tools/ab/$name.so $flags &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
