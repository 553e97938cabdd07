#!/bin/bash
# instruction counts per stage of the step kernel: libraries built with -DARCLE_STOP_AT=k (the wave returns after stage k;
# 0 = the whole step) run the C3 mix under a PMC pass; differences between consecutive rows are the stages' costs
R=${GRAFT_REPO_ROOT:-$(pwd)}
for k in 1 2 3 4 5 0; do
  echo "== ARCLE_STOP_AT=$k"
  ARCLE_HIP_LIB=$R/gpurun_lib_stop$k.so bash $R/tools/gpu_oppmc.sh ${1:--1} 2>&1 | tail -2
done
