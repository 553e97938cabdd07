#!/bin/bash
# Round 4, call A (gpurun from the repo root): smoke, the GPU suite incl. tests/test_round4_hip.py, the bench line in the driver's form and
# in the default form (compact last line; the full record lands in gpurun_out/bench_full_*.json).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short --durations=12 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
echo "== bench (driver form)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.log 2> $O/bench_k20.err; echo "bench rc=$?"; tail -c 4000 $O/bench_k20.log
echo "== bench (default)"; timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -c 4000 $O/bench.log
grep -v BENCH_FULL $O/bench.err | tail -5
