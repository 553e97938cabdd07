#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
summ() { python -c "
import json,sys
for line in sys.stdin:
    if not line.startswith('{'): continue
    d=json.loads(line); r=d['roofline']
    print('%s n_gpus %d value %.1f M/s  ms/step %.4f  kernel %.2f us  achieved %.0f GB/s frac %.3f  regions %d' % (d['config']['id'], d['n_gpus'], d['value']/1e6, d['ms_per_step'], r['avg_launch_us'], r['achieved'], r['frac'], d['timing']['regions']))
    for k in ('floodfill','extras'):
        if k in d: print('   ', k, json.dumps(d[k])[:300])
    if 'cpu_baseline' in d:
        c=d['cpu_baseline']; print('    cpu: C 1T %.2f M/s, all(%d) %.1f M/s | numpy 1P %.1f k/s, all(%d) %.1f k/s' % (c['value']/1e6, c['all_cores']['cores'], c['all_cores']['value']/1e6, c['numpy_step']['value']/1e3, c['numpy_step']['all_cores']['cores'], c['numpy_step']['all_cores']['value']/1e3))
"; }
echo "== default"; timeout 900 python bench.py 2>&1 | tee $O/bench_default.log | summ
echo "== driver args"; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | summ
for c in c2 c4 c5; do echo "== $c"; timeout 600 python bench.py --config $c --no-cpu-baseline 2>&1 | tee $O/bench_$c.log | summ; done
echo "== --gpus 2 on this box (self-spawn; gloo control plane when ranks share a GPU)"; timeout 600 python bench.py --gpus 2 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | tail -3 | summ
echo "== c4 --gpus 2"; timeout 600 python bench.py --gpus 2 --config c4 --steps 50 --warmup 5 2>&1 | tail -3 | summ
