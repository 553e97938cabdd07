#!/bin/bash
# round 3, first GPU pass: smoke, GPU parity tests, bench, ingress forms
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log
echo "== ingress forms"; timeout 300 python tools/maskbench.py > $O/maskbench.log 2>&1; cat $O/maskbench.log | tail -8
echo "== flat rows"; timeout 300 python tools/flatbench.py > $O/flatbench.log 2>&1; tail -8 $O/flatbench.log
