#!/bin/bash
# Round 6, call B: state of the tree on the GPU (full GPU suite, smoke), big-grid step time per op family (where the 40x40 / 64x64 time
# goes), the research step with the SGPR cap raised (gpurun_lib_sgpr96.so = -DARCLE_SGPR_CAP=96), the bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 --tb=short > $O/r6b_pytest_gpu.log 2>&1; echo "rc=$?"; tail -4 $O/r6b_pytest_gpu.log
echo "== big grids by op family (16384 envs)"
for ops in "" 0-9 10-19 20-23 24-27 28-29 30-30 31-33 34-34; do
  timeout 300 python tools/bigbench.py --sizes 40x40,64x64 --envs 16384 ${ops:+--ops $ops} 2>&1 | grep envs
done | tee $O/r6b_big_by_op.txt
echo "== research step, SGPR cap 80 (shipped) / 96"
for rep in 1 2; do
  timeout 300 python tools/researchbench.py 2>&1 | grep -v amdgpu.ids | tail -1
  ARCLE_HIP_LIB=$R/gpurun_lib_sgpr96.so timeout 300 python tools/researchbench.py 2>&1 | grep -v amdgpu.ids | tail -1
done | tee $O/r6b_research_sgpr.txt
echo "== bench"; timeout 900 python bench.py > $O/r6b_bench.json 2>$O/r6b_bench.err; echo "rc=$?"; tail -c 900 $O/r6b_bench.json
