#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s3
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 300 python3 -m pytest tests/test_mt_gpu.py -x -q > $O/pytest_mt.log 2>&1 ); echo "pytest mt rc=$?" | tee -a $O/pytest_mt.log
tail -15 $O/pytest_mt.log
b() { n=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" --no-cpu > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; tail -2 $O/bench_$n.err; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_kernel-parity -o s -- python3 $R/bench.py --workload kernel --rng parity --steps 50 --warmup 5 --no-cpu > $O/kstats_kernel-parity.json 2> $O/kstats_kernel-parity.log
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log 2>&1
head -6 $O/kernel-parity_kernel_stats.csv | cut -c1-160
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O -name "*agent_info.csv" -delete 2>/dev/null || true
