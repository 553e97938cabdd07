#!/bin/bash
# quick iteration on the big-grid translation unit: tools/mk_big.sh <name> [extra flags] -> gpurun_lib_<name>.so from the working tree's
# arcle_big.hip, linked with /tmp/hip_main.o (arcle_hip.hip compiled once: hipcc ... -c arcle_amd/csrc/arcle_hip.hip -o /tmp/hip_main.o)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -mllvm -amdgpu-atomic-optimizer-strategy=DPP "$@" -c $R/arcle_amd/csrc/arcle_big.hip -o /tmp/big_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $R/gpurun_lib_$name.so /tmp/hip_main.o /tmp/big_$name.o
ls -la $R/gpurun_lib_$name.so
