#!/bin/bash
# round 5, big grids: parity + threads-per-workgroup sweep of the workgroup-per-env step kernel (tools/bigbench.py)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_big_hip.py -x -q 2>&1 | tail -5 > gpurun_out/r5n_tests.log
for t in 0 64 128 256; do
  echo "== ARCLE_BIG_THREADS=$t (0 = the library's choice)" >> gpurun_out/r5n.log
  ARCLE_BIG_THREADS=$t python tools/bigbench.py --envs 1024,4096,16384 2>&1 | grep envs >> gpurun_out/r5n.log
done
cat gpurun_out/r5n_tests.log gpurun_out/r5n.log
