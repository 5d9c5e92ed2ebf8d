#!/bin/bash
# Same-box A/B of the one-launch step's placement knobs (diagnostic build: GLB_FIN_LAG = how far behind its row's stats
# blocks a finishing block is dealt, percent of the chip's wave slots, -1 = end of grid; GLB_LDS_PAD = bytes of unused
# LDS per one-wave workgroup = a cap on the waves a CU holds), then in-kernel stamps of the chosen forms.
# Output: gpurun_out/ab/.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/ab
mkdir -p $O
export GLB_DBG_LIB=libglb_hip_dbg.so
for round in 1 2; do
  for lag in -1 25 50 100 200; do
    echo "== round $round GLB_FIN_LAG=$lag"
    GLB_FIN_LAG=$lag python3 $R/tools/kbench.py --iters 200 --quick 2>&1 | grep -E "mask=3" | cut -c1-140
  done
done > $O/ab_lag.log 2>&1
for round in 1 2; do
  for pad in 0 10240 13312 20480; do
    echo "== round $round GLB_LDS_PAD=$pad (fin lag default)"
    GLB_LDS_PAD=$pad python3 $R/tools/kbench.py --iters 200 --quick 2>&1 | grep -E "mask=3" | cut -c1-140
  done
done > $O/ab_pad.log 2>&1
for lag in -1 100; do
  for shape in gpt2 llama; do
    echo "== stamps $shape GLB_FIN_LAG=$lag"
    GLB_FIN_LAG=$lag python3 $R/tools/dbg/stamps.py $shape 12
  done
done > $O/stamps.log 2>&1
cat $O/ab_lag.log $O/ab_pad.log
