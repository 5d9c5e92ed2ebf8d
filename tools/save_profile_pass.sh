#!/bin/bash
# Copy the judged summaries of one tools/profile_round.sh pass (gpurun_out/prof) into profiles/<round>/ with a version
# suffix: tools/save_profile_pass.sh r04 v2
set -e
cd "$(dirname "$0")/.."
R=$1; V=$2; O=gpurun_out/${3:-prof}; D=profiles/$R
mkdir -p $D
for f in $O/bench_*.json; do b=$(basename $f .json); grep "^{" $f > $D/${b}_$V.json || true; done
for f in $O/*_kernel_stats.csv $O/*_pmc_traffic.json $O/*_gpu_busy.txt $O/*_step_table.txt $O/*_request_counters.json $O/rccl_info.log; do
  [ -f "$f" ] || continue
  b=$(basename $f); cp $f $D/${b%.*}_$V.${b##*.}
done
for f in lsm_by_shape.txt trie_by_shape.txt gbench.log; do [ -f $O/$f ] && cp $O/$f $D/${f%.*}_$V.${f##*.}; done
[ -f $O/kstats_lsm.log ] && grep -v "^\[" $O/kstats_lsm.log | tail -12 > $D/lsm_bench_$V.log || true
[ -f $O/kstats_trie.log ] && grep -v "^\[" $O/kstats_trie.log | tail -12 > $D/trie_bench_$V.log || true
ls $D | wc -l
