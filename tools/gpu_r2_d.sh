#!/bin/bash
# in-box A/B (interleaved rounds) of kernel knobs on the fast build: ARCLE_ENVS_PER_WAVE x ARCLE_PF_DIST
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for r in 1 2; do
for EPW in 1 2; do for D in 0 128 256 512; do
  echo -n "round $r EPW=$EPW PF=$D: "
  ARCLE_PF_DIST=$D ARCLE_ENVS_PER_WAVE=$EPW ARCLE_HIP_LIB=$R/gpurun_lib_fast.so timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 20 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f M/s  kernel %.2f us  frac %.3f' % (d['value']/1e6, r['avg_launch_us'], r['frac']))"
done; done; done
