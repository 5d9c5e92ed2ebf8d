#!/bin/bash
# same-box comparison of library builds on the log-softmax micro-benchmark: tools/ab_lsm_lib.sh <rounds> <lib under tools/dbg> ...
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
n=$1; shift
for r in $(seq 1 $n); do
  for lib in "$@"; do
    echo "-- $lib"
    GLB_DBG_LIB=$lib python3 $R/tools/kbench_lsm.py 2>&1 | grep log_softmax | cut -c1-110
  done
done
