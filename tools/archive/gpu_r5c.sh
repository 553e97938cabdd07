#!/bin/bash
# Round 5, call C: the generalised self-ordering launch (full library): GPU tests, then bench legs with the table form (ARCLE_GROUP_OVER_ORDER=0:
# hints / arcle_step_many keep their order tables) against grouping everywhere (=1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== pytest round5 (default env)"; timeout 900 python -m pytest tests/test_round5_hip.py -m gpu -q -x --tb=short > $O/r5c_pytest5.log 2>&1; echo "rc=$?"; tail -4 $O/r5c_pytest5.log
echo "== pytest gpu, table form kept (ARCLE_GROUP_OVER_ORDER=0)"; ARCLE_GROUP_OVER_ORDER=0 timeout 1500 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_round5_hip.py > $O/r5c_pytest.log 2>&1; echo "rc=$?"; tail -6 $O/r5c_pytest.log
for r in 1 2; do for oo in 0 1; do
  echo "== round $r ARCLE_GROUP_OVER_ORDER=$oo"
  ARCLE_GROUP_OVER_ORDER=$oo timeout 900 python bench.py --no-cpu-baseline 2>$O/r5c_err_$oo.log | tail -1 > $O/r5c_bench_${oo}_$r.json
  python - $O/r5c_bench_${oo}_$r.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('value %.1f M/s kernel %.3f us frac %.3f' % (d['value']/1e6, r['avg_launch_us'], r['frac']))
print(' ordered', json.dumps(d.get('ordered')))
print(' legs', json.dumps(d.get('legs_us_per_step')))
print(' ooc', json.dumps(r.get('frac_out_of_cache')))
PY
done; done
