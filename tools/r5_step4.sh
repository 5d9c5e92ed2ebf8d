#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s4
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
b() { n=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" --no-cpu > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; tail -2 $O/bench_$n.err; }
b sis-llama --workload sis-llama --steps 30 --warmup 5
b sis-llama_particlekv --workload sis-llama --particle-kv --steps 30 --warmup 5
b sis-llama_particlekv_resample --workload sis-llama --particle-kv --resample --steps 30 --warmup 5
b sis --workload sis --steps 50 --warmup 10
b sis_particlekv_resample --workload sis --particle-kv --resample --steps 30 --warmup 5
b api_autokv --workload api --auto-kv --steps 30 --warmup 5
b api_autokv_device --workload api --auto-kv --device-batch --steps 30 --warmup 5
b api_device --workload api --device-batch --steps 20 --warmup 3
b trie_selected --workload trie --trie-out selected --steps 50 --warmup 5
b trie_slots --workload trie --trie-out slots --steps 50 --warmup 5
for t in "sis-llama" "sis-llama --particle-kv" "sis --particle-kv"; do
  n=$(echo $t | tr -d ' ' | sed 's/--/_/g')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_$n -o s -- python3 $R/bench.py --workload $t --steps 30 --warmup 5 --no-cpu --no-kv-line > $O/kstats_$n.json 2> $O/kstats_$n.log
  tr=$(find $O/kstats_$n -name "*kernel_trace.csv" | head -1)
  if [ -n "$tr" ]; then
    python3 $R/tools/gpu_busy.py $tr > $O/${n}_gpu_busy.txt 2>&1
    python3 $R/tools/gemm_table.py $tr $O/kstats_$n.json > $O/${n}_step_table.txt 2>&1
  fi
  echo "prof $n done"
done
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log 2>&1
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O -name "*agent_info.csv" -delete 2>/dev/null || true
ls $O | head -5
