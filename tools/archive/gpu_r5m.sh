#!/bin/bash
# Round 5, call M: c5 — flood-fill passes per convergence ballot (fast-build libraries gpurun_lib_u*.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for r in 1 2 3; do for u in 1 2 3; do
  echo -n "round $r unroll $u: "; R=$R ARCLE_HIP_LIB=$R/gpurun_lib_u$u.so timeout 300 python tools/c5exp.py 2>&1 | grep "c5 flags" | tr '\n' ' '; echo
done; done
