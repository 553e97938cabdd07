#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: smoke, GPU parity tests, bench, rocprofv3 kernel trace, PMC passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rocm-smi --showproductname > $O/gpu_info.log 2>&1; nproc >> $O/gpu_info.log; lscpu | head -20 >> $O/gpu_info.log
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py > $O/bench.log 2>&1; echo "bench rc=$?"; grep '^{' $O/bench.log
echo "== rocprof kernel trace of the same command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
python $R/tools/prof_summary.py $O/prof/step_results.db | head -8
echo "== PMC passes"
bash $R/tools/gpu_pmc.sh 2>&1 | tee $O/pmc_summary.txt | tail -40
cd $R && python tools/make_pmc_json.py gpurun_out round2 > $O/pmc_json.log 2>&1; cp profiles/pmc_latest.json $O/pmc_latest.json
echo "== wave trace"
[ -f $R/gpurun_lib_trace.so ] && ARCLE_HIP_LIB=$R/gpurun_lib_trace.so python $R/tools/wavetrace.py > $O/wavetrace.txt 2>&1; tail -12 $O/wavetrace.txt
echo "== other bench configs"
for c in c2 c4 c5; do timeout 600 python $R/bench.py --config $c --no-cpu-baseline > $O/bench_$c.log 2>&1; grep '^{' $O/bench_$c.log | cut -c1-400; done
timeout 600 python $R/bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_gpus2_shared.log 2>&1; grep '^{' $O/bench_gpus2_shared.log | cut -c1-300
echo "== sweep"
for N in 32768 131072; do timeout 600 python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 --envs-per-gpu $N > $O/bench_n$N.log 2>&1; grep '^{' $O/bench_n$N.log | cut -c1-330; done
