#!/bin/bash
# round 5, big grids: the step kernel by class of operation (tools/bigbench.py --ops) and with 512 / 1024-thread workgroups at 127 x 127
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_big_hip.py -x -q 2>&1 | tail -3 > gpurun_out/r5o.log
python tools/bigbench.py --envs 1024,4096 2>&1 | grep envs >> gpurun_out/r5o.log
for t in 256 512; do
  echo "== ARCLE_BIG_THREADS=$t" >> gpurun_out/r5o.log
  ARCLE_BIG_THREADS=$t python tools/bigbench.py --sizes 127x127 --envs 1024,4096 2>&1 | grep envs >> gpurun_out/r5o.log
done
for ops in 0-9 10-19 20-23 24-27 28-29 30-30 31-33 34-34; do
  python tools/bigbench.py --sizes 40x40,127x127 --envs 1024 --ops $ops 2>&1 | grep envs >> gpurun_out/r5o.log
done
cat gpurun_out/r5o.log
