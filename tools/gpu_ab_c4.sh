#!/bin/bash
# interleaved in-box A/B of library builds on the c4 leg (8192 envs, step + fused packed row): us per launch, unhinted / hinted
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cat > /tmp/c4leg.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["R"])
import bench
r = bench.other_configs_leg(torch.device("cuda:0"), K=100)
print("c4 %.2f us  hinted %.2f us | c2 %.2f | c5 %.2f" % (r["c4"]["us_per_step_batch"], r["c4"].get("hinted_us_per_step_batch", 0), r["c2"]["us_per_step_batch"], r["c5"]["us_per_step_batch"]))
PY
for r in 1 2 3; do for lib in "$@"; do echo -n "round $r $lib: "; R=$R ARCLE_HIP_LIB=$R/$lib timeout 300 python /tmp/c4leg.py 2>/dev/null | tail -1; done; done
