#!/bin/bash
# round 4: -DARCLE_FAST_BUILD libraries of the work tree for the streaming-regime A/B (store / load cache policy, speculative grid load)
set -e
R=$(cd $(dirname $0)/.. && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DARCLE_FAST_BUILD -mllvm -amdgpu-kernarg-preload-count=13"
b() { name=$1; shift; hipcc $F "$@" -o $R/gpurun_lib_$name.so $R/arcle_amd/csrc/arcle_hip.hip 2>/dev/null & }
b sA_spec_sc1
b sB_spec_nt '-DARCLE_STREAM_STORE_POLICY="nt"'
b sG_spec_sc1nt '-DARCLE_STREAM_STORE_POLICY="sc1 nt"'
b sH_spec_ent_sc1 -DARCLE_STREAM_EARLY_NT=1
b sI_spec_ent_sc1nt -DARCLE_STREAM_EARLY_NT=1 '-DARCLE_STREAM_STORE_POLICY="sc1 nt"'
b sJ_spec_ent_nt -DARCLE_STREAM_EARLY_NT=1 '-DARCLE_STREAM_STORE_POLICY="nt"'
wait
ls -la $R/gpurun_lib_s*.so
