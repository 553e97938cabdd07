#!/bin/bash
# Round 4, streaming regime A/B (gpurun from the repo root): the headline kernel over batch sizes, one library, the policy forced through
# ARCLE_STREAM_POLICY (0 = the resident-regime kernel; A spec + sc1 stores; B spec + nt stores; H nt-spec + sc1 stores; J nt-spec + nt stores)
# and the workgroup size through ARCLE_WPW.  us per launch by HIP events (bench.py, 60-step regions, distinct action batches).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out; mkdir -p $O
LIB=${LIB:-gpurun_lib_fast.so}
SIZES=${SIZES:-"32768 131072 524288"}
POLICIES=${POLICIES:-"0 A B H J"}
run() {  # N, policy, wpw
  ARCLE_HIP_LIB=$R/$LIB ARCLE_STREAM_POLICY=$2 ARCLE_WPW=$3 timeout 300 python bench.py --no-cpu-baseline --no-extras --no-ordered --steps ${STEPS:-60} --warmup 10 --regions 12 ${EXTRA:-} \
    --envs-per-gpu $1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%8.2f us  frac %.3f  by-traffic %.3f' % (r['avg_launch_us'], r['frac'], r['frac_by_traffic']))"
}
for N in $SIZES; do
  echo "== N=$N"
  for round in 1 2; do
    for P in $POLICIES; do for W in ${WPWS:-0}; do
      echo -n "  r$round policy=$P wpw=$W : "; run $N $P $W
    done; done
  done
done 2>&1 | tee -a $O/r4_stream_ab2.txt
