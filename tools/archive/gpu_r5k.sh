#!/bin/bash
# Round 5, call K: final library — GPU suite, then the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -m gpu -q --maxfail=10 --tb=short --durations=4 2>&1 | tail -9
timeout 900 python bench.py --no-cpu-baseline > $O/bench_k.log 2> $O/bench_k.err; tail -c 2600 $O/bench_k.log
