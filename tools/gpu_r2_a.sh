#!/bin/bash
# round 2, call A: floor measurements (membench) + plane-stride A/B of the current kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
rocm-smi --showproductname > $O/gpu_info.log 2>&1; nproc >> $O/gpu_info.log
echo "== membench"; timeout 900 ./gpurun_membench 300 > $O/membench.log 2>&1; echo "rc=$?"; head -8 $O/membench.log
for PS in 912 1024; do
  for N in 8192 32768 131072; do
    K=400; [ $N -gt 8192 ] && K=100
    echo "== bench PS=$PS N=$N"
    ARCLE_PLANE_STRIDE=$PS timeout 600 python bench.py --no-cpu-baseline --steps $K --warmup 20 --envs-per-gpu $N 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f M/s  kernel %.2f us  achieved %.0f GB/s  frac %.3f' % (d['value']/1e6, r['avg_launch_us'], r['achieved'], r['frac']))"
  done
done
echo "== parity with PS=1024"
ARCLE_PLANE_STRIDE=1024 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or vs_oracle_o2arc or full_size" 2>&1 | tail -3
