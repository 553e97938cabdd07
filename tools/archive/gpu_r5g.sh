#!/bin/bash
# Round 5, call G: rollout kernel compiled for 4 / 5 / 6 waves per SIMD (fast-build libraries gpurun_lib_rw*.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for r in 1 2; do for wv in 4 5 6; do
  echo "== round $r waves/SIMD target $wv"; ARCLE_HIP_LIB=$R/gpurun_lib_rw$wv.so timeout 300 python tools/rolloutbench.py 2>&1 | grep -v amdgpu.ids
done; done
