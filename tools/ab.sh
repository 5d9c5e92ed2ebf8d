#!/bin/bash
# same-box comparison of builds of the library on the fused step: tools/ab.sh <rounds> <libA.so> <libB.so> ...
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
n=$1; shift
for r in $(seq 1 $n); do
  for lib in "$@"; do
    echo "-- $lib"
    GLB_DBG_LIB=$lib python3 $R/tools/kbench.py --iters 200 --quick 2>&1 | grep -E "mask=3" | cut -c1-120
  done
done
