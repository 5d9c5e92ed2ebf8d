#!/bin/bash
# The trie part of tools/profile_round.sh alone (kernel statistics + the two PMC passes of tools/tbench.py): refreshes
# gpurun_out/prof/{trie_kernel_stats.csv,trie_pmc_traffic.json,trie_by_shape.txt,kstats_trie.log} after a change to the trie kernels.
set -e
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp
export TMPDIR=/tmp
O=$R/gpurun_out/prof
mkdir -p $O
rm -rf $O/kstats_trie $O/pmc_fetch_trie $O/pmc_write_trie
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats_trie -o t -- python3 $R/tools/tbench.py > $O/kstats_trie.log 2>&1
python3 $R/tools/trace_by_grid.py $(find $O/kstats_trie -name "*kernel_trace.csv" | head -1) trie_ > $O/trie_by_shape.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_trie -o f -- python3 $R/tools/tbench.py > $O/pmc_f_trie.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_trie -o w -- python3 $R/tools/tbench.py > $O/pmc_w_trie.log 2>&1
python3 $R/tools/pmc_summary.py $O > $O/pmc_summary.log
find $O -name "*.db" -delete 2>/dev/null || true
find $O -name "*kernel_trace.csv" -delete 2>/dev/null || true
ls $O | grep trie
