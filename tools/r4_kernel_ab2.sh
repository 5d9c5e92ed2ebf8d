#!/bin/bash
# Same-box A/B, second set: short chunks last (GLB_SHORT_LAST), and what bounds the steady state (GLB_DBG_MODE 1 = a
# sixteenth of the exponentials: the memory side alone; 2 = rows out of the caches: the issue side alone).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/ab
mkdir -p $O
export GLB_DBG_LIB=libglb_hip_dbg.so
for round in 1 2 3; do
  for sl in 0 1; do
    echo "== round $round GLB_SHORT_LAST=$sl"
    GLB_SHORT_LAST=$sl python3 $R/tools/kbench.py --iters 200 --quick 2>&1 | grep -E "mask=3|mask=0 rng=1" | cut -c1-140
  done
done > $O/ab_short_last.log 2>&1
for mode in 0 1 2; do
  echo "== GLB_DBG_MODE=$mode (0 = the real kernel; 1 = 1/16 of the exponentials; 2 = rows 0..7 only: cache-served)"
  GLB_DBG_MODE=$mode python3 $R/tools/kbench.py --iters 200 --quick 2>&1 | grep -E "mask=3" | cut -c1-140
done > $O/ab_bound.log 2>&1
for sl in 0 1; do
  for shape in gpt2 llama; do
    echo "== stamps $shape GLB_SHORT_LAST=$sl"
    GLB_SHORT_LAST=$sl python3 $R/tools/dbg/stamps.py $shape 12
  done
done > $O/stamps_short_last.log 2>&1
cat $O/ab_short_last.log $O/ab_bound.log
grep -E "==|last stats|last token|finish work mean|lifetime" $O/stamps_short_last.log
