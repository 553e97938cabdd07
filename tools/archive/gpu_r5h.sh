#!/bin/bash
# Round 5, call H: the whole GPU suite + smoke on the final library, then rollout bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -22 $O/pytest_gpu.log
echo "== rollout"; timeout 300 python tools/rolloutbench.py 2>&1 | grep -v amdgpu.ids
