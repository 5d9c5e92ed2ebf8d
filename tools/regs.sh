#!/bin/bash
# register / spill report of one row-kernel translation unit: tools/regs.sh <dt> <mode> [name filter]
cd "$(dirname "$0")/../genlm-backend_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGLB_DT=$1 -DGLB_MODE=$2 \
  -Rpass-analysis=kernel-resource-usage -c glb_row_tu.hip -o /tmp/regs_$1_$2.o 2>&1 |
  grep -A12 "Function Name: .*${3:-row_kernel}" | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize" |
  sed 's/.*remark: //;s/\[-Rpass.*//'
