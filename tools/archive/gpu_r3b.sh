#!/bin/bash
# round 3: GPU parity tests + the bench line with all extras
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== pytest gpu"; timeout 1800 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"; tail -5 $O/bench.err; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/bench.log") if l.startswith("{")][-1])
    print("value", d["value"], "us", d["ms_per_step"] * 1e3, "frac", d["roofline"]["frac"], "frac_by_traffic", d["roofline"]["frac_by_traffic"])
    for k, v in d.get("extras", {}).items():
        if isinstance(v, dict):
            print(" ", k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list))})
            for kk, vv in v.items():
                if isinstance(vv, dict) and "us_per_step_batch" in vv:
                    print("      ", kk, round(vv["us_per_step_batch"], 2), "us", "frac", round(vv.get("roofline", {}).get("frac", 0), 3))
            if "roofline" in v:
                print("      roofline frac", round(v["roofline"]["frac"], 3), "by traffic", v["roofline"].get("frac_by_traffic"))
        else:
            for e in v:
                print(" ", k, e["envs"], round(e["us_per_step_batch"], 2), "us frac", round(e["roofline"]["frac"], 3), "traffic frac", round(e["roofline"]["frac_by_traffic"], 3))
except Exception as exc:
    print("parse failed", exc)
PY
