#!/bin/bash
# interleaved in-box A/B of the libraries given as arguments (bench.py, C3 mix, N=8192): kernel us per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    echo -n "round $r $lib: "
    ARCLE_HIP_LIB=$R/$lib timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 20 ${BENCH_ARGS:-} 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.1f M/s  kernel %.2f us  frac %.3f' % (d['value']/1e6, r['avg_launch_us'], r['frac']))"
  done
done
