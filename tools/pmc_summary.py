"""Condense rocprofv3 outputs under <dir> into the files committed under profiles/: per-kernel stats CSVs
(first lines) and the HBM traffic of the fused kernel per launch (FETCH_SIZE x2 on gfx950, see
MI355X_MICROARCH.md; counters are in KB)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def first(pattern):
    g = sorted(glob.glob(os.path.join(root, pattern), recursive=True))
    return g[0] if g else None


out = {}
for tag, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = first(f"{sub}/**/*counter_collection.csv")
    if not f:
        continue
    per = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != tag:
                continue
            name = row["Kernel_Name"].split("<")[0].replace("void ", "")
            if not name.startswith("glb::"):
                continue
            per.setdefault(name, []).append(float(row["Counter_Value"]))
    for k, v in per.items():
        out[f"{k}:{tag}"] = {"launches": len(v), "mean_raw_KB": sum(v) / len(v)}
fetch = sum(v["mean_raw_KB"] for k, v in out.items() if k.endswith("FETCH_SIZE") and "row_kernel" in k)
write = sum(v["mean_raw_KB"] for k, v in out.items() if k.endswith("WRITE_SIZE") and "row_kernel" in k)
out["hbm_read_bytes_per_launch_corrected"] = fetch * 1024 * 2
out["hbm_write_bytes_per_launch"] = write * 1024
out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --workload kernel "
               "--steps 20 --warmup 2 --no-cpu`; FETCH_SIZE doubled per MI355X_MICROARCH.md; algorithmic bytes per launch = "
               "205873432")
json.dump(out, open(os.path.join(root, "kernel_workload_pmc_traffic.json"), "w"), indent=1)
for tag, sub in (("kernel", "kstats"), ("sis", "sstats")):
    f = first(f"{sub}/**/*kernel_stats.csv")
    if f:
        lines = open(f).read().splitlines()
        keep = [lines[0]] + [ln for ln in lines[1:] if len(ln) < 600][:25]
        open(os.path.join(root, f"{tag}_workload_kernel_stats.csv"), "w").write("\n".join(keep) + "\n")
print(json.dumps(out, indent=1))
