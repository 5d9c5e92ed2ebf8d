#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s9
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?" | tee -a $O/pytest.log
tail -2 $O/pytest.log
b() { n=$1; shift; timeout -k 10 500 python3 $R/bench.py "$@" > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; }
t0=$(date +%s); b default; echo "default bench wall: $(( $(date +%s) - t0 )) s"
b sis-llama_parity --workload sis-llama --rng parity --steps 20 --warmup 3 --no-cpu
b sis-llama_particlekv_parity --workload sis-llama --particle-kv --rng parity --steps 20 --warmup 3 --no-cpu
b rehearse2 --gpus 2 --rehearse-one-gpu --steps 10 --warmup 2 --no-cpu
( cd $R && python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -1
for f in $O/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"]), round(d["ms_per_step"],3), r.get("frac"), d.get("ms_per_step_kv"), d.get("rccl_ranks"), d.get("gloo_ranks"), "cpu" in str(d.get("cpu_baseline"))[:5] or bool(d.get("cpu_baseline")))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
wc -l $O/bench_default.json
