#!/bin/bash
# round 5 evidence: bench lines (default and the driver's form), rocprofv3 kernel trace of the same command (timed regions alone),
# kernel trace with the extras, PMC passes (instruction mix, FETCH_SIZE / WRITE_SIZE in separate passes) -> profiles/pmc_latest.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== bench (default)"; timeout 900 python bench.py > $O/round5_bench.json 2>$O/bench.err; echo "rc=$?"; tail -c 700 $O/round5_bench.json
grep '^BENCH_FULL ' $O/bench.err | cut -c12- > $O/round5_bench_full.json
echo "== bench (driver form)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round5_bench_k20.json 2>$O/bench_k20.err; tail -c 500 $O/round5_bench_k20.json
echo "== rocprof kernel trace of the same command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-extras > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
(python $R/tools/prof_summary.py $O/prof/step_results.db --timed 400 5; tail -1 $O/rocprof.log | cut -c1-1200) > $O/round5_bench_kernel_trace.txt 2>&1; head -14 $O/round5_bench_kernel_trace.txt
echo "== kernel trace of the bench with extras"
rm -rf $O/prof2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o ex -- python $R/bench.py --no-cpu-baseline > $O/rocprof2.log 2>&1; echo "rocprof rc=$?"
python $R/tools/prof_summary.py $O/prof2/ex_results.db > $O/round5_extras_kernel_trace.txt 2>&1; head -40 $O/round5_extras_kernel_trace.txt
echo "== PMC passes"
bash $R/tools/gpu_pmc.sh 2>&1 | tee $O/round5_pmc_summary.txt | tail -40
cd $R && python tools/make_pmc_json.py gpurun_out round5 > $O/pmc_json.log 2>&1; cp profiles/pmc_latest.json $O/pmc_latest.json; cat $O/pmc_latest.json
echo "== big grids (workgroup-per-env kernels): bench + kernel trace"
(python $R/tools/bigbench.py --envs 1024,4096,16384 2>&1 | grep envs) > $O/round5_big_grid.txt
rm -rf $O/prof3
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof3 -o big -- python $R/tools/bigbench.py --sizes 64x64 --envs 4096 > $O/rocprof3.log 2>&1; echo "rocprof rc=$?"
(echo; echo "rocprofv3 --kernel-trace --stats of: python tools/bigbench.py --sizes 64x64 --envs 4096"; python $R/tools/prof_summary.py $O/prof3/big_results.db) >> $O/round5_big_grid.txt 2>&1; tail -12 $O/round5_big_grid.txt
cd $R
echo "== soak (6000 steps)"; SOAK_STEPS=6000 timeout 1200 python tools/soak.py > $O/round5_soak.txt 2>&1; tail -12 $O/round5_soak.txt
