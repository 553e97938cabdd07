#!/bin/bash
# development A/B: gpurun_lib_0base.so from the last commit, gpurun_lib_1new.so from the work tree, both -DARCLE_FAST_BUILD (30 s)
set -e
R=$(cd $(dirname $0)/.. && pwd)
rm -rf /tmp/ab_base && mkdir -p /tmp/ab_base
git -C $R archive HEAD arcle_amd/csrc include | tar -x -C /tmp/ab_base
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DARCLE_FAST_BUILD -mllvm -amdgpu-kernarg-preload-count=13"
hipcc $F -o $R/gpurun_lib_0base.so /tmp/ab_base/arcle_amd/csrc/arcle_hip.hip 2>/dev/null &
hipcc $F "$@" -o $R/gpurun_lib_1new.so $R/arcle_amd/csrc/arcle_hip.hip 2>/dev/null &
wait
ls -la $R/gpurun_lib_0base.so $R/gpurun_lib_1new.so
