#!/bin/bash
# rocprofv3 kernel durations of the fused step on the two headline shapes, once per line of stdin "NAME ENV=VAL ..."
# (tuning knobs: MASK, RNG).  Run on the GPU box through gpurun; output gpurun_out/pp/.
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=$R/gpurun_out/pp
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
while read -r name envs; do
  [ -z "$name" ] && continue
  for shape in gpt2 llama; do
    for e in $envs; do export "$e"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_$shape -o k -- python3 $R/tools/kprof.py $shape ${MASK:-3} ${RNG:-1} 40 > $O/${name}_$shape.log 2>&1 || exit 1
    for e in $envs; do unset "${e%%=*}"; done
    echo "== $name $shape"
    python3 - "$O/${name}_$shape/k_kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "glb::" in r["Name"] and "mask_prepare" not in r["Name"]]
for r in rows:
    print("   %-60s %8.2f us x %s" % (r["Name"].split("(")[0][:60], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("   total %.2f us" % (sum(float(r["AverageNs"]) for r in rows) / 1e3))
PY
  done
done
