#!/bin/bash
# (GPU box) bench lines with the recorded GEMM solutions against the library's defaults, same box, alternating
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
one() { timeout -k 10 600 python bench.py "$@" --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(f\"{d['ms_per_step']:8.3f} ms  {d['value']:10.0f} p/s\" + (f\"   kv {d['ms_per_step_kv']:.3f} ms {d['value_kv']:.0f}\" if 'value_kv' in d else '') + '   ' + d['config'].get('gemms', '')[:60])"; }
for w in "" "--particle-kv --no-kv-line" "--workload sis-llama --steps 20 --warmup 10" "--workload sis-llama --particle-kv --steps 30 --warmup 10" "--workload sis-llama --llama 3-8b --particle-kv --steps 10 --warmup 5" "--workload api --auto-kv --device-batch --steps 20 --warmup 10"; do
  for g in recorded library recorded library; do echo -n "[$w] $g: "; one $w --gemms $g; done
done
