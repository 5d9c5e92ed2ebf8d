#!/bin/bash
# same-box comparison of log-softmax kernels on the diagnostic build: tools/ab_lsm.sh <rounds> <mode>/<variant> ...
# mode (GLB_LSM_GRID): -1 = independent waves, 0 = re-streaming workgroup per row, 1 = resident rows, one workgroup per
# CU, 4 = resident rows, one row per workgroup; variant (GLB_LSM_VARIANT, waves only): waves per SIMD * 10 + store form
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
n=$1; shift
for r in $(seq 1 $n); do
  for g in "$@"; do
    echo "-- mode/variant $g"
    GLB_LSM_GRID=${g%/*} GLB_LSM_VARIANT=${g#*/} GLB_DBG_LIB=libglb_hip_dbg.so python3 $R/tools/kbench_lsm.py 2>&1 | grep log_softmax | cut -c1-100
  done
done
