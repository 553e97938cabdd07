#!/bin/bash
# rocprofv3 kernel traces of the c4 and c5 bench legs (summaries land in gpurun_out/, copied to profiles/ by hand)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in c4 c5 c2; do
  rm -rf $O/prof_$c
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o t -- python $R/bench.py --config $c --no-cpu-baseline > $O/rocprof_$c.log 2>&1; echo "$c rc=$?"
  python $R/tools/prof_summary.py $O/prof_$c/t_results.db --timed 400 5 > $O/kernel_trace_$c.txt; grep '^{' $O/rocprof_$c.log | cut -c1-600 >> $O/kernel_trace_$c.txt; head -5 $O/kernel_trace_$c.txt; grep "timed regions" $O/kernel_trace_$c.txt
done
