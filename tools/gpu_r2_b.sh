#!/bin/bash
# round 2, call B: parity of the lean kernel + bench + per-op PMC
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== parity"; timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -5
for N in 8192 32768 131072; do
  K=400; [ $N -gt 8192 ] && K=100
  echo "== bench N=$N"
  timeout 600 python bench.py --no-cpu-baseline --steps $K --warmup 20 --envs-per-gpu $N 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f M/s  kernel %.2f us  achieved %.0f GB/s  frac %.3f  B/env %.0f' % (d['value']/1e6, r['avg_launch_us'], r['achieved'], r['frac'], r['algorithmic_bytes_per_env_step']))"
done
bash tools/gpu_oppmc.sh "0,10,20,24,26,28,30,31,32,34,-1" 2>&1 | tail -14
bash tools/gpu_opprof.sh 2>&1 | tail -12
