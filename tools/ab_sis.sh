#!/bin/bash
# same-box comparison of library builds inside the SIS loop: tools/ab_sis.sh <rounds> <libA.so> <libB.so> ...
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
n=$1; shift
for r in $(seq 1 $n); do
  for lib in "$@"; do
    GLB_DBG_LIB=$lib python3 $R/tools/bench_with_lib.py --workload ${WL:-sis} --steps 40 --warmup 10 --no-cpu 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib'.ljust(24), 'frac %.3f med %.3f outer %.3f us %.2f ms/step %.2f' % (r['frac'], r['frac_median'], r['frac_outer_events'], r['us_per_launch_mean'], d['ms_per_step']))"
  done
done
