#!/bin/bash
# quick profile pass (gpurun): rocprofv3 kernel stats of the kernel / kernel-llama / sis workloads + bench lines
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r3/prof
rm -rf $O && mkdir -p $O
for w in kernel kernel-llama sis; do
  timeout -k 10 300 python3 $R/bench.py --workload $w --steps 50 --warmup 10 --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$w -o k -- python3 $R/bench.py --workload $w --steps 50 --warmup 5 --no-cpu > $O/ks_$w.log 2>&1
  echo "== $w"; python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('value %.0f ms/step %.3f frac %.3f us mean %.2f median %.2f bytes %.1f MB' % (d['value'], d['ms_per_step'], r.get('frac',0), r.get('us_per_launch_mean',0), r.get('us_per_launch_median',0), r.get('bytes_per_launch',0)/1e6))
except Exception as e: print('bench parse failed', e)
"
  grep -E "glb::" $O/ks_$w/k_kernel_stats.csv | cut -c1-60,60-200 | awk -F'",' '{print substr($1,1,70), $2}' | head -8
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
