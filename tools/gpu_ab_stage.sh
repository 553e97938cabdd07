#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in $R/gpurun_lib_*.so; do
  rm -rf $R/gpurun_out/prof_stage
  ARCLE_HIP_LIB=$lib rocprofv3 --kernel-trace -d $R/gpurun_out/prof_stage -o st -- python $R/tools/stagebench.py > /dev/null 2>&1
  echo "=== $(basename $lib): stages 1,2,3,5,4,0"
  python $R/tools/prof_summary.py $R/gpurun_out/prof_stage/st_results.db --chunk 400 | grep -E "step launches" | awk '{printf "%s ", $6} END {print ""}'
done
