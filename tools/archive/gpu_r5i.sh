#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for wt in "8 24" "48 24" "96 48" "192 96"; do set -- $wt
  ARCLE_HIP_LIB=$R/gpurun_lib_fast.so ARCLE_AUTOTUNE_WARM=$1 ARCLE_AUTOTUNE_TIMED=$2 timeout 300 python tools/autotunebench.py 32768,65536,131072 2>&1 | grep -v amdgpu.ids
done
