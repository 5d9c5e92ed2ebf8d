#!/bin/bash
# Runs on the GPU box (through gpurun): bench lines of every workload, rocprofv3 kernel statistics of the kernel and SIS
# workloads and the HBM traffic counters of the fused step (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
# Output: gpurun_out/prof/ (copy what is to be judged into profiles/rNN/).
set -e
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
for w in kernel kernel-llama; do
  python3 $R/bench.py --workload $w --steps 200 --warmup 10 > $O/bench_$w.json 2> $O/bench_$w.err
done
python3 $R/bench.py --workload sis --steps 50 --warmup 10 > $O/bench_sis.json 2> $O/bench_sis.err
python3 $R/bench.py --workload sis-llama --steps 30 --warmup 5 > $O/bench_sis-llama.json 2> $O/bench_sis-llama.err
for w in "api" "api-coro" "api-logprobs"; do
  python3 $R/bench.py --workload $w --steps 20 --warmup 3 --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err
done
python3 $R/bench.py --workload api --auto-kv --steps 20 --warmup 3 --no-cpu > $O/bench_api_autokv.json 2>> $O/bench_api.err
python3 $R/bench.py --workload api-coro --auto-kv --steps 20 --warmup 3 --no-cpu > $O/bench_api-coro_autokv.json 2>> $O/bench_api.err
python3 $R/bench.py --workload sis --prefix-kv --prompts 8 --steps 30 --warmup 5 --no-cpu > $O/bench_sis_prefixkv_k8.json 2>> $O/bench_sis.err
python3 $R/bench.py --workload sis --prefix-kv --prompts 64 --steps 30 --warmup 5 --no-cpu > $O/bench_sis_prefixkv_k64.json 2>> $O/bench_sis.err
python3 $R/bench.py --workload sis --particle-kv --steps 30 --warmup 5 --no-cpu > $O/bench_sis_particlekv.json 2>> $O/bench_sis.err
python3 $R/bench.py --workload sis --particle-kv --resample --steps 30 --warmup 5 --no-cpu > $O/bench_sis_particlekv_resample.json 2>> $O/bench_sis.err
for w in kernel kernel-llama; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_$w -o k -- python3 $R/bench.py --workload $w --steps 100 --warmup 5 --no-cpu > $O/kstats_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -o f -- python3 $R/bench.py --workload $w --steps 20 --warmup 2 --no-cpu > $O/pmc_f_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -o w -- python3 $R/bench.py --workload $w --steps 20 --warmup 2 --no-cpu > $O/pmc_w_$w.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_sis -o s -- python3 $R/bench.py --workload sis --steps 30 --warmup 3 --no-cpu > $O/kstats_sis.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_sis -o f -- python3 $R/bench.py --workload sis --steps 20 --warmup 0 --no-cpu > $O/pmc_f_sis.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_sis -o w -- python3 $R/bench.py --workload sis --steps 20 --warmup 0 --no-cpu > $O/pmc_w_sis.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_sis-llama -o s -- python3 $R/bench.py --workload sis-llama --steps 20 --warmup 3 --no-cpu > $O/kstats_sis-llama.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_lsm -o l -- python3 $R/tools/kbench_lsm.py > $O/kstats_lsm.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_lsm -o f -- python3 $R/tools/kbench_lsm.py > $O/pmc_f_lsm.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_lsm -o w -- python3 $R/tools/kbench_lsm.py > $O/pmc_w_lsm.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_trie -o t -- python3 $R/tools/tbench.py > $O/kstats_trie.log 2>&1
python3 $R/tools/trace_by_grid.py $(find $O/kstats_lsm -name "*kernel_trace.csv" | head -1) > $O/lsm_by_shape.txt 2>&1
python3 $R/tools/trace_by_grid.py $(find $O/kstats_trie -name "*kernel_trace.csv" | head -1) trie_ > $O/trie_by_shape.txt 2>&1
# ---- round 4: per-particle masks, the verbatim README loop, the KV step, SQ counters, trie traffic -------------------------
python3 $R/bench.py --workload kernel --per-row-masks --steps 200 --warmup 10 --no-cpu > $O/bench_kernel_rowmasks.json 2>> $O/bench_kernel.err
python3 $R/bench.py --workload kernel-llama --per-row-masks --steps 200 --warmup 10 --no-cpu > $O/bench_kernel-llama_rowmasks.json 2>> $O/bench_kernel.err
python3 $R/bench.py --workload sis --per-row-masks --steps 30 --warmup 5 --no-cpu > $O/bench_sis_rowmasks.json 2>> $O/bench_sis.err
python3 $R/bench.py --workload api-readme --steps 10 --warmup 2 --no-cpu > $O/bench_api-readme.json 2>> $O/bench_api.err
# the coroutine workloads with AsyncAmdLM.gather instead of asyncio.gather
python3 $R/bench.py --workload api-coro --llm-gather --steps 20 --warmup 3 --no-cpu > $O/bench_api-coro_llmgather.json 2>> $O/bench_api.err
python3 $R/bench.py --workload api-coro --auto-kv --llm-gather --steps 20 --warmup 3 --no-cpu > $O/bench_api-coro_autokv_llmgather.json 2>> $O/bench_api.err
python3 $R/bench.py --workload api-readme --llm-gather --steps 10 --warmup 2 --no-cpu > $O/bench_api-readme_llmgather.json 2>> $O/bench_api.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_kernel-rowmasks -o k -- python3 $R/bench.py --workload kernel --per-row-masks --steps 100 --warmup 5 --no-cpu > $O/kstats_kernel-rowmasks.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_kernel-rowmasks -o f -- python3 $R/bench.py --workload kernel --per-row-masks --steps 20 --warmup 2 --no-cpu > $O/pmc_f_kernel-rowmasks.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_kernel-rowmasks -o w -- python3 $R/bench.py --workload kernel --per-row-masks --steps 20 --warmup 2 --no-cpu > $O/pmc_w_kernel-rowmasks.log 2>&1
for t in "sis --particle-kv" "api --auto-kv"; do
  n=$(echo $t | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_$n -o s -- python3 $R/bench.py --workload $t --steps 50 --warmup 10 --no-cpu > $O/kstats_$n.log 2>&1
  tr=$(find $O/kstats_$n -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python3 $R/tools/gpu_busy.py $tr > $O/${n}_gpu_busy.txt 2>&1
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_trie -o f -- python3 $R/tools/tbench.py > $O/pmc_f_trie.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_trie -o w -- python3 $R/tools/tbench.py > $O/pmc_w_trie.log 2>&1
for o in rows slots selected; do
  python3 $R/bench.py --workload trie --trie-out $o --steps 50 --warmup 5 $([ $o = rows ] || echo --no-cpu) > $O/bench_trie_$o.json 2>> $O/bench_trie.err
done
python3 $R/tools/gbench.py > $O/gbench.log 2>&1
# ---- round 5: per-step tables of the default lines (GEMM us / TFLOP/s / % of step) from the traces made above
for n in sis sis-llama; do
  tr=$(find $O/kstats_$n -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python3 $R/tools/gemm_table.py $tr $O/bench_$n.json > $O/${n}_step_table.txt 2>&1
done
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
ls $O
