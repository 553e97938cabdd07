#!/bin/bash
# Round 6, call G: the mask ingest of the big-grid kernels on whole words — the library before it (gpurun_lib_premask.so) against the tree's,
# same run: us per step of the C3 mix with the rectangles sent as int8 masks / bit-packed masks; then the big-grid GPU tests and the new tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for rep in 1 2; do for lib in $R/gpurun_lib_premask.so ""; do for ing in mask bits bbox; do
  echo -n "lib ${lib:-tree} : "
  ARCLE_HIP_LIB=$lib timeout 600 python tools/bigbench.py --sizes 40x40,64x64 --envs 16384 --steps 12 --ingress $ing 2>&1 | grep envs | cut -c1-90 | tr '\n' '|'; echo
done; done; done | tee $O/r6g_mask.txt
timeout 900 python -m pytest tests/test_big_hip.py tests/test_round5_hip.py -q -m gpu --tb=short 2>&1 | tail -5
