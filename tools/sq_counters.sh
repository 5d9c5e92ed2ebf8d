#!/bin/bash
# SQ / GRBM counters of the fused step on the two headline shapes (extra bench.py arguments, e.g. --contract poly, as $@): separate --pmc passes (8 SQ slots per
# pass; no trace domains beside --pmc), condensed by tools/sq_summary.py.  Output: gpurun_out/sq/.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/sq
rm -rf $O && mkdir -p $O
rocprofv3 -L > $O/counters_available.txt 2>&1 || true
PASSES=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
        "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
        "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
        "SQ_LEVEL_WAVES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_TRANS")
for w in kernel kernel-llama; do
  i=0
  for p in "${PASSES[@]}"; do
    i=$((i + 1))
    rocprofv3 --pmc $p --output-format csv -d $O/${w}_p$i -o c -- python3 $R/bench.py --workload $w --steps 20 --warmup 2 --no-cpu "$@" > $O/${w}_p$i.log 2>&1 || echo "pass $i of $w failed" >> $O/failed.txt
  done
done
python3 $R/tools/sq_summary.py $O > $O/sq_summary.log 2>&1
find $O -name "*.db" -delete 2>/dev/null || true
cat $O/sq_summary.log | head -80
