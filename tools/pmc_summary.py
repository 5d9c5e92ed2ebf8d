"""Condense the rocprofv3 outputs of tools/profile_round.sh under <dir> into the files committed under profiles/:
per-kernel stats CSVs (first lines) and, per workload, the HBM traffic of one fused call = fused_step_kernel (the few
calls too small for it: chunk_stats_small_kernel + finish_kernel) (FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md
prescribes; WRITE_SIZE as read; counters are in KB)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
ALGO = {"kernel": 1024 * 50257 * 4 + 2 * 1571 * 4 + 1024 * 8, "kernel-llama": 512 * 128256 * 2 + 2 * 4008 * 4 + 512 * 8,
        "kernel-rowmasks": 1024 * 50257 * 4 + 1024 * 1571 * 4 + 1024 * 8, "sis": None}  # sis: 1 launch in 10 has one unique row; mean algorithmic bytes are in the bench line (about 184.8 MB)
KERNELS = ("glb::fused_step_kernel", "glb::chunk_stats_small_kernel", "glb::chunk_stats_kernel", "glb::finish_kernel",
           "glb::mask_prepare_kernel")
CALL_HEADS = ("glb::fused_step_kernel", "glb::chunk_stats_small_kernel", "glb::chunk_stats_kernel")  # one per fused call


def first(pattern):
    g = sorted(glob.glob(os.path.join(root, pattern), recursive=True))
    return g[0] if g else None


for wl in ("kernel", "kernel-llama", "kernel-rowmasks", "sis"):
    out = {}
    for tag, sub in (("FETCH_SIZE", f"pmc_fetch_{wl}"), ("WRITE_SIZE", f"pmc_write_{wl}")):
        f = first(f"{sub}/**/*counter_collection.csv")
        if not f:
            continue
        per = {}
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != tag:
                    continue
                name = row["Kernel_Name"].split("<")[0].replace("void ", "")
                if name in KERNELS:
                    per.setdefault(name, []).append(float(row["Counter_Value"]))
        n_calls = sum(len(v) for k, v in per.items() if k in CALL_HEADS)
        for k, v in per.items():  # mean per FUSED CALL (the sis workload's set-up pass of 10 steps is included: same mix)
            out[f"{k}:{tag}"] = {"launches": len(v), "mean_raw_KB": sum(v) / max(n_calls, 1)}
    if not out:
        continue
    fetch = sum(v["mean_raw_KB"] for k, v in out.items() if k.endswith("FETCH_SIZE"))
    write = sum(v["mean_raw_KB"] for k, v in out.items() if k.endswith("WRITE_SIZE"))
    out["hbm_read_bytes_per_launch_corrected"] = fetch * 1024 * 2
    out["hbm_write_bytes_per_launch"] = write * 1024
    out["algorithmic_bytes_per_launch"] = ALGO[wl]
    out["note"] = (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --workload {wl.replace('-rowmasks', ' --per-row-masks')} "
                   f"--steps 20 --warmup {0 if wl == 'sis' else 2} --no-cpu`; one fused call = fused_step_kernel (small calls: chunk_stats_small_kernel + finish_kernel); FETCH_SIZE doubled "
                   "per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)")
    json.dump(out, open(os.path.join(root, f"{wl}_pmc_traffic.json"), "w"), indent=1)
    print(wl, json.dumps(out, indent=1))
# log-softmax: traffic per launch of logprob_rows_waves_kernel by grid size (tools/kbench_lsm.py sweeps five shapes)
lsm = {}
for tag, sub in (("FETCH_SIZE", "pmc_fetch_lsm"), ("WRITE_SIZE", "pmc_write_lsm")):
    f = first(f"{sub}/**/*counter_collection.csv")
    if not f:
        continue
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == tag and "logprob_rows_waves_kernel" in row["Kernel_Name"]:
                key = f"{row['Kernel_Name'].split('(')[0].replace('void ', '')} grid {row.get('Grid_Size') or row.get('Grid_Size_X')}"
                lsm.setdefault(key, {}).setdefault(tag, []).append(float(row["Counter_Value"]))
if lsm:
    out = {}
    for key, d in sorted(lsm.items()):
        rd = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [0])), 1) * 1024 * 2
        wr = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [0])), 1) * 1024
        out[key] = {"launches": len(d.get("FETCH_SIZE", [])), "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr}
    out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 tools/kbench_lsm.py`; grid = rows x "
                   "chunks x 64 threads: 851968 = 1024 x 50257 (fp32: 205.9 MB in, 205.9 MB out; bf16: 102.9 MB in), 1048576 = 512 x "
                   "128256 bf16 (131.3 MB in, 262.7 MB out); FETCH_SIZE doubled per MI355X_MICROARCH.md")
    json.dump(out, open(os.path.join(root, "lsm_pmc_traffic.json"), "w"), indent=1)
    print("lsm", json.dumps(out, indent=1))
# trie masses: traffic per launch of every trie kernel (tools/tbench.py)
trie = {}
for tag, sub in (("FETCH_SIZE", "pmc_fetch_trie"), ("WRITE_SIZE", "pmc_write_trie")):
    f = first(f"{sub}/**/*counter_collection.csv")
    if not f:
        continue
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") == tag and "trie_" in row["Kernel_Name"]:
                import re

                name = re.search(r"trie_\w+(<[^>]*>)?", row["Kernel_Name"]).group(0)
                key = f"{name} grid {row.get('Grid_Size') or row.get('Grid_Size_X')}"
                trie.setdefault(key, {}).setdefault(tag, []).append(float(row["Counter_Value"]))
if trie:
    out = {}
    for key, d in sorted(trie.items()):
        rd = sum(d.get("FETCH_SIZE", [0])) / max(len(d.get("FETCH_SIZE", [0])), 1) * 1024 * 2
        wr = sum(d.get("WRITE_SIZE", [0])) / max(len(d.get("WRITE_SIZE", [0])), 1) * 1024
        out[key] = {"launches": len(d.get("FETCH_SIZE", [])), "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr}
    out["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 tools/tbench.py`; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (calibrated there for 16-byte-per-lane streaming reads; the level kernels read 16 bytes per lane)")
    json.dump(out, open(os.path.join(root, "trie_pmc_traffic.json"), "w"), indent=1)
    print("trie", json.dumps(out, indent=1))
# the trie bench lines (bench.py --workload trie --trie-out X): HBM traffic of one call = every trie_* launch of the call
for out_kind in ("rows", "slots", "selected", "rowsel", "rowsel-root"):
    wl = f"trie-{out_kind}"
    per_tag = {}
    for tag, sub in (("FETCH_SIZE", f"pmc_fetch_{wl}"), ("WRITE_SIZE", f"pmc_write_{wl}")):
        f = first(f"{sub}/**/*counter_collection.csv")
        if not f:
            continue
        tot, calls = 0.0, 0
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != tag or "trie_" not in row["Kernel_Name"]:
                    continue
                tot += float(row["Counter_Value"])
                if "trie_sweep_kernel" in row["Kernel_Name"] or (
                        "trie_rows_kernel" in row["Kernel_Name"] and ", false," in row["Kernel_Name"].replace("<", ", ").replace(">", ",")):
                    calls += 1  # (one launch over the parts per call; the top, when the plan has one, is a second launch)
        per_tag[tag] = (tot, calls)
    if len(per_tag) == 2:
        out = {"hbm_read_bytes_per_launch_corrected": per_tag["FETCH_SIZE"][0] / max(per_tag["FETCH_SIZE"][1], 1) * 1024 * 2,
               "hbm_write_bytes_per_launch": per_tag["WRITE_SIZE"][0] / max(per_tag["WRITE_SIZE"][1], 1) * 1024,
               "calls": per_tag["FETCH_SIZE"][1],
               "note": (f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --workload trie --trie-out "
                        f"{out_kind} --steps 20 --warmup 2 --no-cpu`; per call = all trie_* launches of the call; FETCH_SIZE doubled per "
                        "MI355X_MICROARCH.md")}
        json.dump(out, open(os.path.join(root, f"{wl}_pmc_traffic.json"), "w"), indent=1)
        print(wl, json.dumps(out, indent=1))
# request counters of the trie kernel (TCP / TCC: is glb_trie_rows bound by requests, as DESIGN.md §10 says, or by bytes?)
for f in sorted(glob.glob(os.path.join(root, "pmc_req_trie*", "**", "*counter_collection.csv"), recursive=True)):
    agg = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "trie_rows_kernel" not in row["Kernel_Name"] and "trie_sweep_kernel" not in row["Kernel_Name"]:
                continue
            k = row["Counter_Name"]
            agg.setdefault(k, []).append(float(row["Counter_Value"]))
    if agg:
        tag = os.path.relpath(f, root).split(os.sep)[0][len("pmc_req_"):].split("_")[0]  # pmc_req_<workload>_<counter>
        out = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in sorted(agg.items())}
        prev = {}
        pth = os.path.join(root, f"{tag}_request_counters.json")
        if os.path.exists(pth):
            prev = json.load(open(pth))
        prev.update(out)
        json.dump(prev, open(pth, "w"), indent=1)
        print(tag, json.dumps(out))

def short_name(name, limit=150):
    """Kernel names of Tensile / ATen run to several hundred characters: keep the head that identifies them (macro-tile for
    Tensile, functor for ATen) - the NAME is cut, never the row's numbers (round 4 cut the line and lost them)."""
    return name if len(name) <= limit else name[:limit] + "..."


for f in sorted(glob.glob(os.path.join(root, "kstats_*"))):
    tag = os.path.basename(f)[len("kstats_"):]
    f = first(f"kstats_{tag}/**/*kernel_stats.csv")
    if not f or not os.path.isdir(os.path.join(root, f"kstats_{tag}")):
        continue
    with open(f, newline="") as fh:
        rows = list(csv.reader(fh))
    head, body = rows[0], rows[1:]
    ours = [r for r in body if "glb::" in r[0] or "anonymous namespace)::" in r[0] and "at::native" not in r[0]]
    rest = [r for r in body if r not in ours]
    rest.sort(key=lambda r: -float(r[2]))
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(head)
        for r in ours + rest[:40]:  # this library's kernels: all; the others: the 40 with the most time, numbers intact
            w.writerow([short_name(r[0])] + [float(x) if i != 0 else x for i, x in enumerate(r[1:], 1)])
        if len(rest) > 40:
            tail = rest[40:]
            w.writerow([f"({len(tail)} more kernels)", sum(float(r[1]) for r in tail), sum(float(r[2]) for r in tail), 0.0,
                        sum(float(r[4]) for r in tail), 0.0, 0.0, 0.0])
