#!/bin/bash
# Round 5, call E: arcle_transition_rows with the early pass-through of the planes an op cannot write (1new) against HEAD (0base); fast-build libraries
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for r in 1 2 3; do for lib in gpurun_lib_0base.so gpurun_lib_1new.so; do
  ARCLE_HIP_LIB=$R/$lib timeout 300 python tools/transbench.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
done; done
