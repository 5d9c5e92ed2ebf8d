#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s6
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
b() { n=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" --no-cpu > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; tail -2 $O/bench_$n.err; }
b sis-llama --workload sis-llama --steps 30 --warmup 5
b sis-llama2 --workload sis-llama --steps 30 --warmup 5
b sis --workload sis --steps 50 --warmup 10
b kernel --workload kernel --steps 200 --warmup 10
b kernel-llama --workload kernel-llama --steps 200 --warmup 10
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_sis-llama -o s -- python3 $R/bench.py --workload sis-llama --steps 30 --warmup 5 --no-cpu > $O/kstats_sis-llama.json 2> $O/kstats_sis-llama.log
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log 2>&1
grep short_attention $O/sis-llama_kernel_stats.csv | cut -c1-200
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O -name "*agent_info.csv" -delete 2>/dev/null || true
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_*.json | paste - - - 
