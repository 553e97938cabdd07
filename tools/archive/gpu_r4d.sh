#!/bin/bash
# Round 4, call D: GPU suite (speculative / streaming instantiations vs the oracle), research step with and without resets / hints, bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out; mkdir -p $O
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
{
echo "== research step: episode limit 100 (1 % of the envs auto-reset per step) / 1000000 (none)"
for r in 1 2; do for L in 100 1000000; do echo -n "r$r limit=$L: "; ARCLE_BENCH_RESEARCH_LIMIT=$L timeout 300 python tools/researchbench.py 2>/dev/null | tail -1; done; done
} 2>&1 | tee $O/r4_research.txt
echo "== bench"; timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -c 3800 $O/bench.log
