#!/bin/bash
# PyTorch's TunableOp (GEMM solution selection among the library's own kernels) on the KV-row steps: time and ms per step
cd $GRAFT_REPO_ROOT
for w in "sis-llama --particle-kv" "sis --particle-kv"; do
for e in 0 1; do
  echo "== $w tunableop $e"
  t0=$(date +%s)
  PYTORCH_TUNABLEOP_ENABLED=$e PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable_%d.csv timeout -k 10 500 python bench.py --workload $w --steps 30 --warmup 8 --no-cpu 2>gpurun_out/tun_$e.err > gpurun_out/tun_$e.json
  python -c "import sys,json; d=json.loads(open('gpurun_out/tun_$e.json').read()); print(d['ms_per_step'], d['value'])"
  echo "wall $(( $(date +%s) - t0 )) s"
done
done
wc -l /tmp/tunable_0.csv; cut -c1-160 /tmp/tunable_0.csv | head -30
