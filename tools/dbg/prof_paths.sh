#!/bin/bash
# rocprofv3 kernel durations of the fused step on the two headline shapes, for each launch sequence / workgroup size
# given as "NAME ENV=VAL ..." lines on stdin (run on the GPU box through gpurun; output gpurun_out/pp/)
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
    grep -h "glb::" $O/${name}_$shape/k_kernel_stats.csv | grep -v mask_prepare | cut -d, -f1-4,6,7 | cut -c1-150
  done
done
