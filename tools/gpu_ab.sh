#!/bin/bash
# A/B several builds of the library: per-op GPU kernel durations for each gpurun_lib_*.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in $R/gpurun_lib_*.so; do
  echo "=== $(basename $lib)"
  ARCLE_HIP_LIB=$lib bash $R/tools/gpu_opprof.sh "$@" | grep -v chunk | awk '{printf "%s ", $6} END {print ""}'
done
