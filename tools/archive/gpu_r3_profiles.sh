#!/bin/bash
# round 3 evidence: bench lines, rocprofv3 kernel trace of the same command, PMC passes, FETCH/WRITE_SIZE calibration
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== bench (default)"; timeout 900 python bench.py > $O/round3_bench.json 2>$O/bench.err; echo "rc=$?"; cut -c1-600 $O/round3_bench.json
echo "== bench (driver form)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/round3_bench_k20.json 2>>$O/bench.err; cut -c1-400 $O/round3_bench_k20.json
echo "== rocprof kernel trace of the same command"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python $R/bench.py --no-cpu-baseline --no-extras > $O/rocprof.log 2>&1; echo "rocprof rc=$?"
(python $R/tools/prof_summary.py $O/prof/step_results.db --timed 400 5; grep '^{' $O/rocprof.log | cut -c1-700) > $O/round3_bench_kernel_trace.txt 2>&1; head -12 $O/round3_bench_kernel_trace.txt
echo "== kernel trace of the bench with extras (research / ingress / transition kernels)"
rm -rf $O/prof2
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o ex -- python $R/bench.py --no-cpu-baseline > $O/rocprof2.log 2>&1; echo "rocprof rc=$?"
python $R/tools/prof_summary.py $O/prof2/ex_results.db > $O/round3_extras_kernel_trace.txt 2>&1; head -30 $O/round3_extras_kernel_trace.txt
echo "== PMC passes"
bash $R/tools/gpu_pmc.sh 2>&1 | tee $O/round3_pmc_summary.txt | tail -40
echo "== counter calibration on known byte counts (tools/wcalib.hip)"
for cnt in WRITE_SIZE FETCH_SIZE; do
  for n in 8192 131072; do
    rm -rf $O/cal_${cnt}_$n
    rocprofv3 --pmc $cnt --kernel-trace -d $O/cal_${cnt}_$n -o c --output-format csv -- $R/gpurun_wcalib $n 4 20 > $O/cal_${cnt}_$n.log 2>&1
    python - $O/cal_${cnt}_$n $cnt $n <<'PY'
import sys, csv, glob, collections
d, cnt, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == cnt:
            acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
known = 4 * n * 1024
for k, v in sorted(acc.items()):
    m = sum(v[2:]) / max(len(v[2:]), 1)
    print(f"  {cnt} N={n:6d} {k[:40]:40s} launches {len(v):3d} counter {m:12.1f} KiB = {m * 1024 / known:6.3f} x the {known} bytes moved")
PY
  done
done | tee $O/round3_counter_calibration.txt
