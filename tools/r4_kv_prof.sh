#!/bin/bash
# Where the KV step's milliseconds go: bench lines + rocprofv3 kernel statistics of `sis --particle-kv`
# and `api --auto-kv` (kernel-trace only).  Output: gpurun_out/kvprof/.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/kvprof
rm -rf $O && mkdir -p $O
python3 $R/bench.py --workload sis --particle-kv --steps 50 --warmup 10 --no-cpu > $O/bench_sis_particlekv.json 2> $O/bench_sis_particlekv.err
python3 $R/bench.py --workload api --auto-kv --steps 30 --warmup 5 --no-cpu > $O/bench_api_autokv.json 2> $O/bench_api_autokv.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_sis_particlekv -o s -- python3 $R/bench.py --workload sis --particle-kv --steps 50 --warmup 10 --no-cpu > $O/kstats_sis_particlekv.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_api_autokv -o s -- python3 $R/bench.py --workload api --auto-kv --steps 30 --warmup 5 --no-cpu > $O/kstats_api_autokv.log 2>&1
for t in sis_particlekv api_autokv; do
  f=$(find $O/kstats_$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -45 $f | cut -c1-220 > $O/${t}_kernel_stats.csv
  tr=$(find $O/kstats_$t -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python3 $R/tools/gpu_busy.py $tr > $O/${t}_gpu_busy.txt 2>&1
done
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
cat $O/bench_sis_particlekv.json $O/bench_api_autokv.json | cut -c1-400
cat $O/sis_particlekv_gpu_busy.txt $O/api_autokv_gpu_busy.txt
head -30 $O/sis_particlekv_kernel_stats.csv | cut -c1-160
