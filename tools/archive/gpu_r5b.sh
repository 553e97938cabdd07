#!/bin/bash
# Round 5, call B: group sizes of the self-ordering launch (fast-build libraries gpurun_lib_gs*.so), natural / pre-sorted op streams
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for r in 1 2 3; do
for gs in ${SIZES:-16 32 64}; do
  lib=gpurun_lib_gs$gs.so
  for wpw in 8 4; do
    echo "== round $r $lib wpw=$wpw"
    GRP_L4=1 ARCLE_HIP_LIB=$R/$lib ARCLE_GROUP_WPW=$wpw timeout 300 python tools/grpbench.py 8192 $gs 2>&1 | grep -v "amdgpu.ids"
  done
done
done
for gs in ${SIZES:-16 32 64}; do for n in 4096 16384 32768; do for wpw in 8 4; do
  echo "== gs$gs n=$n wpw $wpw"; GRP_ONLY=natural ARCLE_GROUP_MIN=0 ARCLE_HIP_LIB=$R/gpurun_lib_gs$gs.so ARCLE_GROUP_WPW=$wpw timeout 300 python tools/grpbench.py $n $gs 2>&1 | grep -v "amdgpu.ids"
done; done; done
