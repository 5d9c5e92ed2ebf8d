set -e
mkdir -p gpurun_out/p6n
cd /root/repo
for w in "kernel" "kernel --per-row-masks" "kernel --contract hw --per-row-masks" "kernel-llama" "kernel-llama --per-row-masks" "kernel --contract hw"; do
  n=$(echo "$w" | tr ' ' '_' | tr -d '-')
  python bench.py --workload $w --steps 400 --warmup 20 --no-cpu > gpurun_out/p6n/bench_$n.json 2> gpurun_out/p6n/bench_$n.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/p6n/bench_$n.json").read().strip().splitlines()[-1])
print("$w", d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"))
PY
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "step or raw or mask" > gpurun_out/p6n/pytest.log 2>&1; tail -3 gpurun_out/p6n/pytest.log
