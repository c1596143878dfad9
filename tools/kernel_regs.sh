#!/bin/bash
# VGPR / SGPR / LDS / occupancy of the kernels in one csrc/*.hip whose names match $2.
#   bash tools/kernel_regs.sh dz_rainbow adam
f=dqn_zoo_amd/csrc/$1.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I include \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kr_$1.o 2>&1 | \
  awk -v pat="$2" '/Function Name:/ {name=$0; show = (name ~ pat)} show && /(Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize)/ {sub(/^.*remark: [^ ]* /, ""); print}'
