#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s7
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
b() { n=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" --no-cpu > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; tail -2 $O/bench_$n.err; }
b kernel_rowmasks --workload kernel --per-row-masks --steps 200 --warmup 10
b kernel_rowmasks_churn0 --workload kernel --per-row-masks --mask-churn 0 --steps 200 --warmup 10
b kernel_rowmasks_churn10 --workload kernel --per-row-masks --mask-churn 0.1 --steps 200 --warmup 10
b kernel-llama_rowmasks_churn10 --workload kernel-llama --per-row-masks --mask-churn 0.1 --steps 200 --warmup 10
b sis_rowmasks --workload sis --per-row-masks --steps 30 --warmup 5
b kernel_peaked --workload kernel --logits peaked --steps 200 --warmup 10
b kernel_peaked_eosonly --workload kernel --logits peaked --mask eos-only --steps 200 --warmup 10
b trie_rowsel --workload trie --trie-out rowsel --steps 50 --warmup 5
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*' $O/bench_*.json | paste - - -
ls $O/bench_*.json
