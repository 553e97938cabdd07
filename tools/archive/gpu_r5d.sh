#!/bin/bash
# Round 5, call D: after retiring the table form (ABI 5): the whole GPU suite, smoke, the bench line (driver form + default), autotune reports
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_gpu.log
echo "== bench (driver form)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.log 2> $O/bench_k20.err; echo "bench rc=$?"; tail -c 3500 $O/bench_k20.log
echo "== bench (default, no cpu baseline)"; timeout 900 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -c 3500 $O/bench.log
echo "== autotune"; timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
for n in (4096, 8192, 16384, 32768, 65536, 131072):
    for K in (1, 64):   # the action batch the policy just wrote (cache-resident) / a stream of distinct batches
        bb_np, op_np = bench.make_actions(K, n, 5)
        bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
        b = bench.make_batch(dev, n, seed=11)
        FL = b.elide_flag | 1
        before = b.launch_info("bbox", FL)
        rows = b.autotune("bbox", bb[0], op[0], FL)
        print(f"n={n} table plan {before}  autotune: " + "  ".join(f"{'G' if r['orders_itself'] else r['policy'] or '0'}/{r['waves_per_workgroup']}:{r['us_per_launch']:.2f}" for r in rows), flush=True)
        del b; torch.cuda.empty_cache()
        break
PY
