#!/bin/bash
# quick GPU iteration: a few parity tests + staged kernel timings under rocprofv3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or 30-30 or full_size" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stage
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_stage -o st -- python $R/tools/stagebench.py > /dev/null 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_stage/st_results.db --chunk 400 | grep -E "arcle_step|step launches"
