#!/bin/bash
# builds gpurun_lib_0base.so from the last commit (exported with git archive: the work tree is never touched)
# and gpurun_lib_1new.so from the working tree, for tools/gpu_ab2.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
rm -rf /tmp/ab_base && mkdir -p /tmp/ab_base
git -C $R archive HEAD arcle_amd/csrc include | tar -x -C /tmp/ab_base
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=13 -o $R/gpurun_lib_0base.so /tmp/ab_base/arcle_amd/csrc/arcle_hip.hip 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-kernarg-preload-count=13 "$@" -o $R/gpurun_lib_1new.so $R/arcle_amd/csrc/arcle_hip.hip 2>/dev/null
ls -la $R/gpurun_lib_*.so
