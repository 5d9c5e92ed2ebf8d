#!/bin/bash
# Round 5, the trie's sweep kernel: bench lines, rocprofv3 kernel statistics, HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and
# request counters of the slots and all-nodes outputs, tools/tbench.py's table -> gpurun_out/prof5t (copy what is to be
# judged into profiles/r05 with tools/save_profile_pass.sh).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/prof5t
rm -rf $O; mkdir -p $O
for o in rows slots selected rowsel rowsel-root; do
  python3 $R/bench.py --workload trie --trie-out $o --steps 50 --warmup 5 --no-cpu > $O/bench_trie_$o.json 2> $O/bench_trie_$o.err
done
python3 $R/tools/tbench.py > $O/trie_bench.log 2>&1
for o in rows slots; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_trie-$o -o s -- python3 $R/bench.py --workload trie --trie-out $o --steps 50 --warmup 5 --no-cpu > $O/kstats_trie-$o.json 2> $O/kstats_trie-$o.log
  f=$(find $O/kstats_trie-$o -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/trie-${o}_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_trie-$o -o f -- python3 $R/bench.py --workload trie --trie-out $o --steps 20 --warmup 2 --no-cpu > $O/pmc_f_trie-$o.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_trie-$o -o w -- python3 $R/bench.py --workload trie --trie-out $o --steps 20 --warmup 2 --no-cpu > $O/pmc_w_trie-$o.log 2>&1
done
for c in TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_req_trie-slots_$c -o q -- python3 $R/bench.py --workload trie --trie-out slots --steps 10 --warmup 2 --no-cpu > $O/pmc_q_$c.log 2>&1 || echo "counter $c unavailable"
done
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log 2>&1
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O -name "*agent_info.csv" -delete 2>/dev/null || true
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null || true
ls $O | wc -l
