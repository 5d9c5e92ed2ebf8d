#!/bin/bash
# (GPU box) Record, per GEMM shape of the bench workloads, the fastest of the library's own solutions (PyTorch TunableOp) into
# gpurun_out/tuned_<arch>.csv - copy it to genlm-backend_amd/tuned/<arch>.csv to ship it (genlm_backend_amd.gemm_tuning).
# Usage: tools/record_gemm_tuning.sh [part]   part 1: the default line's shapes (GPT-2-small), 4: the other GPT-2 workloads,
# 2: the Llama-3.2-1B shapes, 3: Llama-3-8B.  A part starts from the shipped file (gpurun_out/ does not travel to the box).
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
F=$R/gpurun_out/tuned_gfx950.csv
[ -f $R/genlm-backend_amd/tuned/gfx950.csv ] && [ ! -f $F ] && cp $R/genlm-backend_amd/tuned/gfx950.csv $F
run() { echo "== $*"; timeout -k 10 900 python bench.py "$@" --gemms tune --gemms-file $F --no-cpu 2>/dev/null | cut -c1-160; grep -c Gemm $F; }
case "${1:-1}" in
1)
  run
  ;;
4)
  run --particle-kv --resample --no-kv-line
  run --prefix-kv --prompts 8 --no-kv-line
  run --prefix-kv --prompts 64 --no-kv-line
  run --workload api --steps 20 --warmup 10
  run --workload api --auto-kv --steps 20 --warmup 10
  run --workload api --auto-kv --device-batch --steps 20 --warmup 10
  run --workload api-logprobs --steps 5 --warmup 2
  ;;
2)
  run --workload sis-llama --steps 20 --warmup 10
  run --workload sis-llama --particle-kv --steps 20 --warmup 10
  run --workload sis-llama --particle-kv --resample --steps 20 --warmup 10
  ;;
3)
  run --workload sis-llama --llama 3-8b --particle-kv --steps 10 --warmup 5
  run --workload sis-llama --llama 3-8b --steps 10 --warmup 10
  ;;
esac
wc -l $F
