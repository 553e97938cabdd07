#!/bin/bash
# Round 5, call L: the late strata of a self-ordering launch request their default env's grid plane speculatively (ARCLE_GROUP_SPEC = first such stratum)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for r in 1 2 3; do for sp in 32 28 24 20 16 8 0; do
  echo -n "round $r spec_from=$sp: "; ARCLE_GROUP_SPEC=$sp GRP_ONLY=natural ARCLE_HIP_LIB=$R/gpurun_lib_spec.so timeout 300 python tools/grpbench.py 8192 32 2>&1 | grep "grouped=1\|rows" | tr '\n' ' '; echo
done; done
