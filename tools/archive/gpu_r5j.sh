#!/bin/bash
# Round 5, call J: the group dealt by trading (two classes) vs by a stable counting sort on 2 / 3 / 4 duration classes (fast-build libraries)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for r in 1 2 3; do
  echo "== round $r trade"; GRP_ONLY=natural ARCLE_HIP_LIB=$R/gpurun_lib_trade.so timeout 300 python tools/grpbench.py 8192 32 2>&1 | grep "grouped=1"
  for c in 2 3 4 r; do
    echo "== round $r sort classes=$c"; ARCLE_GROUP_CLASSES=$c GRP_ONLY=natural ARCLE_HIP_LIB=$R/gpurun_lib_sort.so timeout 300 python tools/grpbench.py 8192 32 2>&1 | grep "grouped=1"
  done
done
