#!/bin/bash
# round 6 evidence: full GPU suite + smoke, bench lines (default and the driver's form), rocprofv3 kernel trace of the same command, kernel
# trace with the extras, PMC passes (instruction mix, FETCH_SIZE / WRITE_SIZE in separate passes) -> profiles/pmc_latest.json, big grids
# (times by size and op family, instruction counts per wave, kernel trace), soak
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/round6_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/round6_smoke.log | cut -c1-200
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=20 --tb=short > $O/round6_pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/round6_pytest_gpu.log | tail -2
echo "== bench (default)"; timeout 900 python bench.py > $O/round6_bench.json 2>$O/bench.err; echo "rc=$?"; tail -c 700 $O/round6_bench.json
grep '^BENCH_FULL ' $O/bench.err | cut -c12- > $O/round6_bench_full.json
echo "== bench (driver form)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_k20.json 2>$O/bench_k20.err; tail -c 500 $O/round6_bench_k20.json
echo "== bench --gpus 2 on one GPU (gloo; c3 + the bounded c4 / c5 legs)"; timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/round6_bench_gpus2_shared.json 2>$O/bench_g2.err; tail -c 900 $O/round6_bench_gpus2_shared.json
echo "== rocprof kernel trace of the same command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-extras > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
(python $R/tools/prof_summary.py $O/prof/step_results.db --timed 400 5; tail -1 $O/rocprof.log | cut -c1-1200) > $O/round6_bench_kernel_trace.txt 2>&1; head -14 $O/round6_bench_kernel_trace.txt
echo "== kernel trace of the bench with extras"
rm -rf $O/prof2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o ex -- python $R/bench.py --no-cpu-baseline > $O/rocprof2.log 2>&1; echo "rocprof rc=$?"
python $R/tools/prof_summary.py $O/prof2/ex_results.db > $O/round6_extras_kernel_trace.txt 2>&1; head -30 $O/round6_extras_kernel_trace.txt
echo "== PMC passes"
bash $R/tools/gpu_pmc.sh 2>&1 | tee $O/round6_pmc_summary.txt | tail -40
cd $R && python tools/make_pmc_json.py gpurun_out round6 > $O/pmc_json.log 2>&1; cp profiles/pmc_latest.json $O/pmc_latest.json; cat $O/pmc_latest.json
echo "== big grids (workgroup-per-env kernels): bench by size, by op family, kernel trace, PMC"
(python $R/tools/bigbench.py --envs 1024,4096,16384 2>&1 | grep envs
 for ops in 0-9 10-19 20-23 24-27 28-29 30-30 31-33 34-34; do python $R/tools/bigbench.py --sizes 40x40,64x64 --envs 16384 --ops $ops 2>&1 | grep envs | cut -c1-120; done) > $O/round6_big_grid.txt
cd /tmp; rm -rf $O/prof3
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof3 -o big -- python $R/tools/bigbench.py --sizes 64x64 --envs 16384 > $O/rocprof3.log 2>&1; echo "rocprof rc=$?"
(echo; echo "rocprofv3 --kernel-trace --stats of: python tools/bigbench.py --sizes 64x64 --envs 16384"; python $R/tools/prof_summary.py $O/prof3/big_results.db) >> $O/round6_big_grid.txt 2>&1; tail -12 $O/round6_big_grid.txt
cd $R
(for cpt in 1 2; do for ops in 0-9 20-23 0-34; do for size in 40x40 64x64 127x127; do
  echo "-- ARCLE_BIG_CPT=$cpt (chunks per thread) ops $ops $size, 4096 envs, 12 eager steps"
  ARCLE_BIG_CPT=$cpt bash tools/gpu_kpmc.sh python $R/tools/bigbench.py --eager --sizes $size --envs 4096 --steps 12 --ops $ops 2>&1 | grep -v "reset"
done; done; done) > $O/round6_big_grid_pmc.txt 2>&1; tail -8 $O/round6_big_grid_pmc.txt
echo "== soak (6000 steps)"; SOAK_STEPS=6000 timeout 1200 python tools/soak.py > $O/round6_soak.txt 2>&1; tail -12 $O/round6_soak.txt
