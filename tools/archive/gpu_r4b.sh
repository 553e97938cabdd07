#!/bin/bash
# Round 4, call B: the GPU suite (hint API, ordered instantiations), the streaming-regime A/B, c2 against its wave count / workgroup size.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out; mkdir -p $O
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --maxfail=40 --tb=short -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
echo "== c2: wave count and workgroup size"
for N in 256 512 1024 2048 4096 8192; do for W in 8 4; do
  echo -n "c2 N=$N wpw=$W: "; ARCLE_WPW=$W timeout 300 python bench.py --config c2 --envs-per-gpu $N --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us' % d['roofline']['avg_launch_us'])"
done; done 2>&1 | tee $O/r4_c2_sweep.txt
echo "== streaming A/B"; bash tools/archive/gpu_r4_stream.sh
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -c 3500 $O/bench.log
