#!/bin/bash
# Round 6, call K: instruction counts per STAGE of the big-grid step (builds with -DARCLE_BIG_STOP_AT=k: the workgroup leaves after stage k)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for ops in 0-9 20-23; do for lib in stop1 stop2 stop3 stop4 tree; do
  [ $lib = tree ] && unset ARCLE_HIP_LIB || export ARCLE_HIP_LIB=$R/gpurun_lib_$lib.so
  echo -n "ops $ops $lib: "; bash tools/gpu_kpmc.sh python $R/tools/bigbench.py --eager --sizes 40x40 --envs 4096 --steps 12 --ops $ops 2>&1 | grep "big_step" | cut -c60-
done; done | tee $O/r6k_stages.txt
