#!/bin/bash
# round 5, big grids after the SWAR passes: threads-per-workgroup sweep again (ARCLE_BIG_THREADS; 0 = the library's choice)
mkdir -p gpurun_out; rm -f gpurun_out/r5s.log
for t in 0 128 256 512; do
  echo "== ARCLE_BIG_THREADS=$t" >> gpurun_out/r5s.log
  ARCLE_BIG_THREADS=$t python tools/bigbench.py --envs 1024,16384 2>&1 | grep envs | cut -c1-75 >> gpurun_out/r5s.log
done
cat gpurun_out/r5s.log
