#!/bin/bash
# interleaved A/B of gpurun_lib_*.so on the C3 mix: R rounds x 300 launches each, median of GPU-side durations
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ROUNDS=${1:-4}
for r in $(seq $ROUNDS); do
  for lib in $R/gpurun_lib_*.so; do
    rm -rf /tmp/prof_ab
    ARCLE_HIP_LIB=$lib rocprofv3 --kernel-trace -d /tmp/prof_ab -o ab -- python $R/tools/opprof.py ${2:--1,-1,-1} > /dev/null 2>&1
    python - /tmp/prof_ab/ab_results.db $(basename $lib) <<'PY'
import sqlite3, sys, numpy as np
c=sqlite3.connect(sys.argv[1])
d=np.array([ (e-s)/1e3 for n,s,e in c.execute("select name,start,end from kernels order by start") if "arcle_step" in n])
print(f"{sys.argv[2]:28s} n={len(d)} median {np.median(d):6.2f}  mean {d.mean():6.2f}  p10 {np.percentile(d,10):6.2f}  min {d.min():6.2f}")
PY
  done
done
