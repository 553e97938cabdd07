#!/bin/bash
# builds an alternative library (both translation units) from the working tree for A/B runs: tools/mk_lib.sh <name> [extra hipcc flags]
# -> gpurun_lib_<name>.so at the repo root (git-ignored, travels with gpurun); run with ARCLE_HIP_LIB=$R/gpurun_lib_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=13 "$@" -c $R/arcle_amd/csrc/arcle_hip.hip -o $T/a.o 2>/dev/null &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -mllvm -amdgpu-atomic-optimizer-strategy=DPP "$@" -c $R/arcle_amd/csrc/arcle_big.hip -o $T/b.o 2>/dev/null &
wait
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $R/gpurun_lib_$name.so $T/a.o $T/b.o
rm -rf $T
ls -la $R/gpurun_lib_$name.so
