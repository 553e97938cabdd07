#!/bin/bash
# bench line + rocprofv3 kernel trace of the same command + the other configs; artefacts land in gpurun_out/ (copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== bench"; timeout 900 python bench.py > $O/bench.log 2>&1; grep '^{' $O/bench.log | cut -c1-260
echo "== driver args"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.log 2>&1; grep '^{' $O/bench_k20.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp; rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
python $R/tools/prof_summary.py $O/prof/step_results.db --timed 400 5 > $O/kernel_trace_summary.txt; grep '^{' $O/rocprof.log >> $O/kernel_trace_summary.txt; head -6 $O/kernel_trace_summary.txt; grep "timed regions" $O/kernel_trace_summary.txt
cd $R
for c in c2 c4 c5; do timeout 600 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.log 2>&1; grep '^{' $O/bench_$c.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['id'], 'value %.1f M/s ms/step %.4f kernel %.2f us achieved %.0f GB/s' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_us'], r['achieved']), d.get('floodfill',''))"; done
for N in 32768 131072; do timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 --envs-per-gpu $N > $O/bench_n$N.log 2>&1; grep '^{' $O/bench_n$N.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('N', d['config']['envs_per_gpu'], 'value %.1f M/s kernel %.2f us achieved %.0f GB/s frac %.3f' % (d['value']/1e6, r['avg_launch_us'], r['achieved'], r['frac']))"; done
timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_gpus2_shared.log 2>&1; grep '^{' $O/bench_gpus2_shared.log | cut -c1-200
timeout 600 python bench.py --gpus 2 --config c4 --steps 20 --warmup 5 --regions 3 > $O/bench_c4_gpus2_shared.log 2>&1; grep '^{' $O/bench_c4_gpus2_shared.log | cut -c1-200
