#!/bin/bash
# round 4 evidence: bench lines (default and the driver's form), rocprofv3 kernel trace of the same command (timed regions alone),
# kernel trace with the extras, PMC passes (instruction mix, FETCH_SIZE / WRITE_SIZE in separate passes) -> profiles/pmc_latest.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== bench (default)"; timeout 900 python bench.py > $O/round4_bench.json 2>$O/bench.err; echo "rc=$?"; tail -c 700 $O/round4_bench.json
grep '^BENCH_FULL ' $O/bench.err | cut -c12- > $O/round4_bench_full.json
echo "== bench (driver form)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round4_bench_k20.json 2>$O/bench_k20.err; tail -c 500 $O/round4_bench_k20.json
echo "== rocprof kernel trace of the same command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-extras > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
(python $R/tools/prof_summary.py $O/prof/step_results.db --timed 400 5; tail -1 $O/rocprof.log | cut -c1-1200) > $O/round4_bench_kernel_trace.txt 2>&1; head -14 $O/round4_bench_kernel_trace.txt
echo "== kernel trace of the bench with extras"
rm -rf $O/prof2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o ex -- python $R/bench.py --no-cpu-baseline > $O/rocprof2.log 2>&1; echo "rocprof rc=$?"
python $R/tools/prof_summary.py $O/prof2/ex_results.db > $O/round4_extras_kernel_trace.txt 2>&1; head -40 $O/round4_extras_kernel_trace.txt
echo "== PMC passes"
bash $R/tools/gpu_pmc.sh 2>&1 | tee $O/round4_pmc_summary.txt | tail -40
cd $R && python tools/make_pmc_json.py gpurun_out round4 > $O/pmc_json.log 2>&1; cp profiles/pmc_latest.json $O/pmc_latest.json; cat $O/pmc_latest.json
echo "== soak (6000 steps)"; SOAK_STEPS=6000 timeout 1200 python tools/soak.py > $O/round4_soak.txt 2>&1; tail -12 $O/round4_soak.txt
