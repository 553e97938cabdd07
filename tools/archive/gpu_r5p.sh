#!/bin/bash
# round 5, arcle_transition_rows: two waves per row (op + pass-through copier) vs one (ARCLE_TRANSITION_SPLIT=0), interleaved; parity first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_round3_hip.py tests/test_features_hip.py -x -q -k "transition or rows" 2>&1 | tail -3 > $O/r5p.log
for r in 1 2 3; do for s in 0 1; do
  (echo -n "split=$s: "; ARCLE_TRANSITION_SPLIT=$s timeout 300 python tools/transbench.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo) >> $O/r5p.log
done; done
cat $O/r5p.log
