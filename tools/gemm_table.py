"""Per-step table of where a workload's GPU time goes, from a rocprofv3 kernel trace and the bench line of the same command:

    python3 tools/gemm_table.py <kernel_trace.csv> <bench line .json> [steps-marker kernel substring]

GEMM = every Tensile kernel (`Cijk_*` / `Custom_Cijk_*`: the transformer body's linear layers and the output embedding - the only
MFMA work on the path).  Their FLOPs are not in the trace; the bench line carries them (`config.gemm_flops_per_step`: 2 x the
linear layers' weights x the tokens fed + 2 x d x V x the rows through the output embedding, summed analytically per step).
A "step" in the trace = one fused call (`fused_step_kernel`, or `chunk_stats_small_kernel` for the calls too small for it):
set-up and warm-up steps run the same mix, so the per-step means hold for the timed region."""
import csv
import json
import sys

trace, line = sys.argv[1], sys.argv[2]
bench = {}
for ln in open(line):
    if ln.startswith("{"):
        bench = json.loads(ln)
groups = {"GEMM (Tensile)": lambda n: n.startswith("Cijk_") or n.startswith("Custom_Cijk_"),
          "glb attention": lambda n: "short_attention" in n or "slab_attention" in n,
          "library attention (attn_fwd / SDPA)": lambda n: "attn_fwd" in n or "fmha" in n.lower(),
          "fused step (glb)": lambda n: "fused_step_kernel" in n or "chunk_stats" in n or "finish_kernel" in n,
          "other glb kernels": lambda n: "glb::" in n or ("anonymous namespace)::" in n and "at::native" not in n),
          "rms_norm / layer_norm": lambda n: "rms_norm" in n.lower() or "layer_norm" in n.lower() or "RowwiseMoments" in n or "vectorized_layer_norm" in n,
          "elementwise / copies (ATen)": lambda n: "at::native" in n}
tot = {k: [0, 0.0] for k in groups}
tot["everything else"] = [0, 0.0]
steps = 0
t_min, t_max = None, None
with open(trace, newline="") as fh:
    for row in csv.DictReader(fh):
        n = row["Kernel_Name"].replace("void ", "")
        b, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
        if "fused_step_kernel" in n or "chunk_stats_small_kernel" in n:
            steps += 1
        for k, pred in groups.items():
            if pred(n):
                break
        else:
            k = "everything else"
        tot[k][0] += 1
        tot[k][1] += (e - b) * 1e-3
busy = sum(v[1] for v in tot.values())
steps = max(steps, 1)
ms_step = bench.get("ms_per_step")
flops = (bench.get("config") or {}).get("gemm_flops_per_step")
print(f"# {trace}\n# bench line: ms_per_step {ms_step}, value {bench.get('value')}; steps in the trace (fused calls): {steps}")
print(f"{'group':40s} {'launches/step':>14s} {'us/step':>10s} {'% of GPU time':>14s} {'% of step (wall)':>17s}")
for k, (cnt, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    wall = f"{us / steps / (ms_step * 1e3) * 100:16.1f}%" if ms_step else " " * 17
    print(f"{k:40s} {cnt / steps:14.1f} {us / steps:10.1f} {us / busy * 100:13.1f}% {wall}")
print(f"{'sum of kernel durations':40s} {'':14s} {busy / steps:10.1f}")
if flops:
    g = tot["GEMM (Tensile)"][1] / steps
    print(f"GEMM: {flops / 1e9:.1f} GFLOP per step (analytic, bench line) / {g:.1f} us = {flops / (g * 1e-6) / 1e12:.1f} TFLOP/s")
