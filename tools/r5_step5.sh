#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s5
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 300 python3 -m pytest tests/test_host_gpu.py -x -q -k "selected_masses or trie" > $O/pytest_trie.log 2>&1 ); echo "pytest rc=$?"; tail -3 $O/pytest_trie.log
for t in "" "--no-speculate"; do
  n=spec$(echo $t | tr -d ' -')
  python3 $R/bench.py --workload sis --particle-kv $t --steps 50 --warmup 10 --no-cpu > $O/bench_$n.json 2> $O/bench_$n.err
  rocprofv3 --kernel-trace --output-format csv -d $O/k_$n -o s -- python3 $R/bench.py --workload sis --particle-kv $t --steps 30 --warmup 5 --no-cpu > $O/k_$n.json 2> $O/k_$n.log
  tr=$(find $O/k_$n -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/gaps.py $tr 14 > $O/gaps_$n.txt 2>&1
  python3 $R/tools/gpu_busy.py $tr >> $O/gaps_$n.txt 2>&1
  cat $O/gaps_$n.txt
done
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O -name "*agent_info.csv" -delete 2>/dev/null || true
grep -h value $O/bench_spec*.json | cut -c1-120
