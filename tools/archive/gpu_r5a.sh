#!/bin/bash
# Round 5, call A: the launch that orders itself (ARCLE_STEPX_GROUPED) — GPU suite for parity, then an interleaved in-box A/B of the headline
# (bench.py, K single-step calls in one graph) with grouping off / on in 8-wave and 4-wave workgroups.  One library, switched by environment.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest gpu (round-4 + hip parity)"; timeout 1200 python -m pytest tests/test_round4_hip.py tests/test_hip_parity.py -m gpu -q -x --tb=short > $O/r5a_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/r5a_pytest.log
one() {  # label, env assignments...
  local label=$1; shift
  echo -n "$label: "
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 20 ${BENCH_ARGS:-} 2>$O/r5a_err.log | grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; o=d.get('ordered') or {}
print('value %.1f M/s  kernel %.3f us  frac %.3f | ordered: %s' % (d['value']/1e6, r['avg_launch_us'], r['frac'], json.dumps({k:o[k] for k in o if 'us' in k or 'value' in k})[:300]))"
}
for r in 1 2 3; do
  one "round $r grouped=0        " ARCLE_GROUPED=0
  one "round $r grouped=1 wpw=8  " ARCLE_GROUPED=1 ARCLE_GROUP_WPW=8
  one "round $r grouped=1 wpw=4  " ARCLE_GROUPED=1 ARCLE_GROUP_WPW=4
done
echo "== fast-build libraries: scalar inputs + second round trip for traded slots (grp) vs all 16 envs' inputs through vector loads (grpvec)"
for r in 1 2 3; do
  for lib in gpurun_lib_grp.so gpurun_lib_grpvec.so; do
    [ -f $R/$lib ] && one "round $r $lib wpw=8" ARCLE_HIP_LIB=$R/$lib ARCLE_GROUP_WPW=8
  done
done
[ -f $R/gpurun_lib_grpvec.so ] && one "grpvec wpw=4" ARCLE_HIP_LIB=$R/gpurun_lib_grpvec.so ARCLE_GROUP_WPW=4
echo "== sizes (grouped off / on, 8-wave)"
for n in 4096 16384 32768; do
  BENCH_ARGS="--envs-per-gpu $n --no-ordered" one "n=$n grouped=0" ARCLE_GROUPED=0
  BENCH_ARGS="--envs-per-gpu $n --no-ordered" one "n=$n grouped=1" ARCLE_GROUPED=1
done
