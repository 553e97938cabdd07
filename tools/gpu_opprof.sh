#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_op
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_op -o op -- python $R/tools/opprof.py "$@" 2>&1 | grep chunk | tr '\n' ' '; echo
python $R/tools/prof_summary.py $R/gpurun_out/prof_op/op_results.db --chunk 100 | grep -E "step launches"
