#!/bin/bash
# A/B of gpurun_lib_*.so in the throughput regime: bench.py at N=131072 envs (kernel ~110 us: launch noise is small)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2 3; do
  for lib in $R/gpurun_lib_*.so; do
    ARCLE_HIP_LIB=$lib python $R/bench.py --envs-per-gpu ${1:-131072} --steps 150 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$(basename $lib)', '%.2f us/launch  %.0f M steps/s  rollout %.0f M/s' % (r['avg_launch_us'], d['value']/1e6, d['extras']['rollout']['value']/1e6))"
  done
done
