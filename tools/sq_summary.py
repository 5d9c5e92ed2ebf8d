"""Condense the rocprofv3 --pmc passes of tools/sq_counters.sh under <dir> into <workload>_sq_counters.json: per kernel of this
library, the mean of every counter per launch (launches of the dominant grid only) and the ratios the guide reads from
them (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_BUSY_CYCLES per SE/XCD instance; GRBM_GUI_ACTIVE summed over the 8 XCDs)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
for wl in ("kernel", "kernel-llama", "kernel-rowmask", "sis"):
    per = {}
    for d in sorted(glob.glob(os.path.join(root, f"{wl}_p*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"]
                    if "glb::" not in name:
                        continue
                    key = (name.split("(")[0].replace("void ", ""), row.get("Grid_Size") or row.get("Grid_Size_X"))
                    per.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not per:
        continue
    out = {}
    for (name, grid), ctrs in sorted(per.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
        m = {c: sum(v) / len(v) for c, v in ctrs.items()}
        m["launches"] = max(len(v) for v in ctrs.values())
        wc = m.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if c in m:
                    m[c + "/SQ_WAVE_CYCLES"] = m[c] / wc
        if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
            m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
        if m.get("SQ_WAVES") and wc:
            m["wave_cycles_x4_per_wave"] = 4 * wc / m["SQ_WAVES"]
        out[f"{name} grid {grid}"] = m
    json.dump(out, open(os.path.join(root, f"{wl}_sq_counters.json"), "w"), indent=1)
    print(wl, json.dumps(out, indent=1))
