#!/bin/bash
# Round-4 workloads (per-particle masks, the verbatim README loop, the default line with value_kv).  Output: gpurun_out/new/.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/new
rm -rf $O && mkdir -p $O
python3 $R/bench.py --steps 30 --warmup 10 > $O/bench_sis_default.json 2> $O/bench_sis_default.err
python3 $R/bench.py --workload kernel --per-row-masks --steps 200 --warmup 10 --no-cpu > $O/bench_kernel_rowmasks.json 2> $O/err1
python3 $R/bench.py --workload kernel-llama --per-row-masks --steps 200 --warmup 10 --no-cpu > $O/bench_kernel-llama_rowmasks.json 2> $O/err2
python3 $R/bench.py --workload sis --per-row-masks --steps 30 --warmup 10 --no-cpu > $O/bench_sis_rowmasks.json 2> $O/err3
python3 $R/bench.py --workload api-readme --steps 10 --warmup 2 --no-cpu > $O/bench_api-readme.json 2> $O/err4
python3 $R/bench.py --workload api --auto-kv --steps 30 --warmup 5 --no-cpu > $O/bench_api_autokv.json 2> $O/err5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_kernel_rowmasks -o k -- python3 $R/bench.py --workload kernel --per-row-masks --steps 100 --warmup 5 --no-cpu > $O/kstats_kernel_rowmasks.log 2>&1
f=$(find $O/kstats_kernel_rowmasks -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-200 > $O/kernel_rowmasks_kernel_stats.csv
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
for f in $O/bench_*.json; do echo "== $f"; python3 - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("unreadable:", e); sys.exit(0)
r = d.get("roofline") or {}
print({k: d.get(k) for k in ("value", "ms_per_step", "value_kv", "ms_per_step_kv")}, {k: r.get(k) for k in ("frac", "us_per_launch_mean", "bytes_per_launch", "frac_outer_events")})
PY
done
cat $O/kernel_rowmasks_kernel_stats.csv
tail -3 $O/err*
