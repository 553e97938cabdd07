#!/bin/bash
# Round 4, call C: small-batch speculation (c2 and the 30x30 kernel at 1024-4096 envs), streaming variants over the size range.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out; mkdir -p $O
q() { tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us' % d['roofline']['avg_launch_us'])"; }
{
echo "== c2 (10x10, generic width): speculative grid load x workgroup size"
for r in 1 2; do for S in 0 4096; do for W in 8 4; do
  echo -n "r$r c2 N=1024 spec_small_max=$S wpw=$W: "; ARCLE_SPEC_SMALL_MAX=$S ARCLE_WPW=$W timeout 300 python bench.py --config c2 --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | q
done; done; done
echo "== c3 kernel at small batches: speculative (streaming instantiation, sc1 stores) on / off"
for r in 1 2; do for N in 1024 2048 4096; do for S in 0 4096; do
  echo -n "r$r c3 N=$N spec_small_max=$S: "; ARCLE_SPEC_SMALL_MAX=$S timeout 300 python bench.py --envs-per-gpu $N --no-cpu-baseline --no-extras --no-ordered --steps 200 --warmup 20 2>/dev/null | q
done; done; done
} 2>&1 | tee $O/r4_small_spec.txt
SIZES="16384 24576 32768 65536 131072 262144" bash tools/archive/gpu_r4_stream.sh
