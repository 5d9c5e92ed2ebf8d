#!/bin/bash
# Round 5 additions to tools/profile_round.sh (run as a second gpurun call: each stays under the 20-minute limit): the parity
# (torch CPU generator on the device) lines, real-shaped logits, the Llama-shaped steps with their per-step GEMM tables and
# GPU-busy, the device-resident API batch, the trie's selected / per-row forms with their HBM traffic and request counters,
# RCCL's own account of the one-rank group.  Output: gpurun_out/prof2/ (tools/save_profile_pass.sh r05 vN prof2 copies the
# judged summaries into profiles/r05/).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/prof2
rm -rf $O && mkdir -p $O
b() { n=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; }
b kernel_parity --workload kernel --rng parity --steps 100 --warmup 5
b kernel-llama_parity --workload kernel-llama --rng parity --steps 100 --warmup 5 --no-cpu
b sis_parity --workload sis --rng parity --steps 30 --warmup 5 --no-cpu
b sis_particlekv_parity --workload sis --particle-kv --rng parity --steps 30 --warmup 5 --no-cpu
b kernel_peaked --workload kernel --logits peaked --steps 200 --warmup 10 --no-cpu
b kernel_eosonly --workload kernel --mask eos-only --steps 200 --warmup 10 --no-cpu
b kernel_peaked_eosonly --workload kernel --logits peaked --mask eos-only --steps 200 --warmup 10 --no-cpu
b kernel-llama_peaked --workload kernel-llama --logits peaked --steps 200 --warmup 10 --no-cpu
b sis-llama_particlekv --workload sis-llama --particle-kv --steps 30 --warmup 5 --no-cpu
b sis-llama_particlekv_resample --workload sis-llama --particle-kv --resample --steps 30 --warmup 5 --no-cpu
b api_autokv_device --workload api --auto-kv --device-batch --steps 30 --warmup 5 --no-cpu
b api_device --workload api --device-batch --steps 20 --warmup 3 --no-cpu
for o in selected rowsel rowsel-root; do b trie_$o --workload trie --trie-out $o --steps 50 --warmup 5 --no-cpu; done
# ---- kernel statistics, GPU-busy and the per-step table (GEMM us / TFLOP/s / % of step) of the step workloads
for t in "sis" "sis-llama" "sis --particle-kv" "sis-llama --particle-kv" "api --auto-kv --device-batch"; do
  n=$(echo $t | tr -d ' ' | sed 's/--/_/g')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_$n -o s -- python3 $R/bench.py --workload $t --steps 30 --warmup 5 --no-cpu --no-kv-line > $O/kstats_$n.json 2> $O/kstats_$n.log
  tr=$(find $O/kstats_$n -name "*kernel_trace.csv" | head -1)
  if [ -n "$tr" ]; then
    python3 $R/tools/gpu_busy.py $tr > $O/${n}_gpu_busy.txt 2>&1
    python3 $R/tools/gaps.py $tr 8 >> $O/${n}_gpu_busy.txt 2>&1
    python3 $R/tools/gemm_table.py $tr $O/kstats_$n.json > $O/${n}_step_table.txt 2>&1
  fi
  echo "prof $n done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_kernel-parity -o s -- python3 $R/bench.py --workload kernel --rng parity --steps 50 --warmup 5 --no-cpu > $O/kstats_kernel-parity.json 2> $O/kstats_kernel-parity.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_kernel-peaked -o s -- python3 $R/bench.py --workload kernel --logits peaked --steps 100 --warmup 5 --no-cpu > $O/kstats_kernel-peaked.json 2> $O/kstats_kernel-peaked.log
# ---- the trie's bench lines: HBM traffic per call, and the request counters of trie_rows_kernel
for o in rows slots selected rowsel; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_trie-$o -o f -- python3 $R/bench.py --workload trie --trie-out $o --steps 20 --warmup 2 --no-cpu > $O/pmc_f_trie-$o.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_trie-$o -o w -- python3 $R/bench.py --workload trie --trie-out $o --steps 20 --warmup 2 --no-cpu > $O/pmc_w_trie-$o.log 2>&1
done
for c in TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_req_trie-rows_$c -o q -- python3 $R/bench.py --workload trie --trie-out rows --steps 10 --warmup 2 --no-cpu > $O/pmc_q_$c.log 2>&1 || echo "counter $c unavailable"
done
# ---- RCCL's own words about the one-rank group and the 4 KiB all-gather
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL python3 $R/tools/rccl_info.py > $O/rccl_info.log 2>&1
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log 2>&1
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O -name "*agent_info.csv" -delete 2>/dev/null || true
find $O -name "*counter_collection.csv" -size +2M -delete 2>/dev/null || true
ls $O | wc -l
