#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics of both bench workloads and the HBM traffic
# counters of the fused kernel (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/prof/
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/prof
rm -rf $O && mkdir -p $O
python3 bench.py --workload kernel --steps 200 --warmup 10 > $O/bench_kernel.json
python3 bench.py --workload sis --steps 50 --warmup 5 > $O/bench_sis.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats -o k -- python3 bench.py --workload kernel --steps 100 --warmup 5 --no-cpu > $O/k_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sstats -o s -- python3 bench.py --workload sis --steps 30 --warmup 3 --no-cpu > $O/s_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --workload kernel --steps 20 --warmup 2 --no-cpu > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 bench.py --workload kernel --steps 20 --warmup 2 --no-cpu > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O
find $O -name "*.db" -delete 2>/dev/null || true
ls -R $O | head -40
