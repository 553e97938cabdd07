#!/bin/bash
# Round 5, call F: the row tail's completion signal (single-env step / transition latency), GPU tests of the single-env classes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest"; timeout 900 python -m pytest tests/test_features_hip.py tests/test_round3_hip.py tests/test_hip_parity.py -m gpu -q -x --tb=short > $O/r5f_pytest.log 2>&1; echo "rc=$?"; tail -4 $O/r5f_pytest.log
for r in 1 2 3; do timeout 300 python tools/singlebench.py 2>&1 | grep -v amdgpu.ids; done
