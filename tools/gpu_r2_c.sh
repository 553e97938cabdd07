#!/bin/bash
# A/B of envs-per-wave on the fast build (bbox / 30x30 / O2ARC table only): bench at several N, then the wave trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for EPW in 1 2; do
  for N in 8192 32768 131072; do
    K=400; [ $N -gt 8192 ] && K=100
    echo -n "EPW=$EPW N=$N: "
    ARCLE_ENVS_PER_WAVE=$EPW ARCLE_HIP_LIB=$R/gpurun_lib_fast.so timeout 600 python bench.py --no-cpu-baseline --no-extras --steps $K --warmup 20 --envs-per-gpu $N 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f M/s  kernel %.2f us  achieved %.0f GB/s  frac %.3f  B/env %.0f' % (d['value']/1e6, r['avg_launch_us'], r['achieved'], r['frac'], r['algorithmic_bytes_per_env_step']))"
  done
done
for EPW in 1 2; do echo "== trace EPW=$EPW"; ARCLE_ENVS_PER_WAVE=$EPW ARCLE_HIP_LIB=$R/gpurun_lib_trace.so python tools/wavetrace.py 2>&1 | tail -12; done
