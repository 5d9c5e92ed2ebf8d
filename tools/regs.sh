#!/bin/bash
# register / spill report of one row-kernel translation unit: tools/regs.sh <dt> [name filter]
cd "$(dirname "$0")/../genlm-backend_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGLB_DT=$1 \
  -Rpass-analysis=kernel-resource-usage -c glb_chunk_tu.hip -o /tmp/regs_$1.o 2>&1 |
  grep -A12 "Function Name: .*${2:-chunk_stats_kernel}" | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize" |
  sed 's/.*remark: //;s/\[-Rpass.*//'
