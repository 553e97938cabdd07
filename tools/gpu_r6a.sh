#!/bin/bash
# Round 6, call A: big-grid step kernel — LEAN instantiations (flags / ingress family / one chunk per thread folded at compile time) against
# the generic kernel (ARCLE_BIG_GENERIC=1), same library, same runs; then the big-grid GPU tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for g in 1 0; do
  echo "== ARCLE_BIG_GENERIC=$g"
  ARCLE_BIG_GENERIC=$g timeout 600 python tools/bigbench.py --sizes 40x40,64x64,127x127 --envs 4096,16384 2>&1 | grep -v amdgpu.ids
done | tee $O/r6a_bigbench.txt
timeout 900 python -m pytest tests/test_big_hip.py -q -m gpu --tb=short 2>&1 | tail -5
