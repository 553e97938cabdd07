#!/bin/bash
# round 5, big grids: no global store ahead of a workgroup barrier (stores deferred behind the step's last barrier) — parity, mix, per class
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_big_hip.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2 > gpurun_out/r5r.log
python tools/bigbench.py --envs 1024,4096,16384 2>&1 | grep envs | cut -c1-100 >> gpurun_out/r5r.log
for ops in 0-9 10-19 20-23 24-27 28-29; do
  python tools/bigbench.py --sizes 40x40,127x127 --envs 1024 --ops $ops 2>&1 | grep envs | cut -c1-100 >> gpurun_out/r5r.log
done
cat gpurun_out/r5r.log
