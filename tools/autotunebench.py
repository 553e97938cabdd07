#!/usr/bin/env python3
"""arcle_autotune against the truth: every launch plan forced through the tuned-plan slot and timed on a graph of K distinct action batches
(what bench.py's batch_sweep leg replays), next to what arcle_autotune itself measured for it.  ARCLE_AUTOTUNE_WARM / _TIMED vary its sample."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); K = 24
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [32768, 65536]
for n in sizes:
    bb_np, op_np = bench.make_actions(K, n, 5)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    b = bench.make_batch(dev, n, seed=11)
    FL = b.elide_flag | 1
    def enqueue(sh):
        for i in range(K):
            b.step_bbox_ptr(bb[i].data_ptr(), op[i].data_ptr(), FL, sh)
    sec0, _ = bench.graph_time(dev, enqueue, K, reps=7, warm=6)
    rows = b.autotune("bbox", bb, op, FL)
    sec1, _ = bench.graph_time(dev, enqueue, K, reps=7, warm=6)
    name = lambda r: ("grouped" if r["orders_itself"] else (r["policy"] or "plain")) + f"/{r['waves_per_workgroup']}w"
    print(f"n={n} warm={os.environ.get('ARCLE_AUTOTUNE_WARM','-')} timed={os.environ.get('ARCLE_AUTOTUNE_TIMED','-')}: table plan graph {sec0*1e6:.2f} us; autotune chose {name(rows[0])} -> graph {sec1*1e6:.2f} us; "
          + "  ".join(f"{name(r)} {r['us_per_launch']:.2f}" for r in rows[:6]), flush=True)
    del b; torch.cuda.empty_cache()
