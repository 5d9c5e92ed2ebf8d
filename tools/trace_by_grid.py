"""Per-(kernel, grid) durations of a rocprofv3 --kernel-trace CSV: tools/trace_by_grid.py <kernel_trace.csv> [name filter]
(kernel stats average over every launch of a kernel; a micro-benchmark that sweeps shapes needs them apart)."""
import collections, csv, statistics, sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else "glb::"
g = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if flt in n:
        short = n.replace("void ", "").replace("(anonymous namespace)::", "")
        short = short[: short.find("(")] if "(" in short else short  # (the argument list goes)
        g[(short, int(r.get("Grid_Size") or r["Grid_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{'kernel':60s} {'grid (threads)':>14s} {'launches':>8s} {'mean us':>9s} {'median':>9s} {'min':>9s} {'max':>9s}")
for (n, grid), v in sorted(g.items()):
    print(f"{n:60s} {grid:14d} {len(v):8d} {statistics.mean(v) / 1e3:9.2f} {statistics.median(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} {max(v) / 1e3:9.2f}")
