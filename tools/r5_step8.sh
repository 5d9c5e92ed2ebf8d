#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/r5s8
rm -rf $O && mkdir -p $O
( cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ); echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
timeout -k 5 120 $R/tools/dbg/valu_rate > $O/valu_rate.log 2>&1; echo "valu_rate rc=$?"
grep -i "pk_fma\|pk_add\| fma \|fma:\|add:\|ldexp\|max" $O/valu_rate.log | head -12
bash $R/tools/r4_sq.sh > $O/sq.log 2>&1; echo "sq rc=$?"
cp $R/gpurun_out/sq/*_sq_counters.json $O/ 2>/dev/null
ls $O
