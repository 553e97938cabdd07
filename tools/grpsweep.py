#!/usr/bin/env python3
"""Self-ordering launches across batch sizes: us per step with ARCLE_GROUPED=0 (the launcher's own choice of kernel / cache policy) and =1
(ARCLE_GROUP_MAX lifted), action stream cache-resident (K distinct batches, K=24) or streaming (K=160)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12288, 16384, 24576, 32768, 49152, 65536, 131072]
os.environ["ARCLE_GROUP_MAX"] = "100000000"
for n in sizes:
    for K in (24, 160):
        bbox_np, op_np = bench.make_actions(K, n, 5)
        bbox, ops = torch.from_numpy(bbox_np).to(dev), torch.from_numpy(op_np).to(dev)
        res = []
        for grouped in ("0", "1", "0", "1"):
            os.environ["ARCLE_GROUPED"] = grouped
            b = bench.make_batch(dev, n, seed=11)
            FL = b.elide_flag | bench.STEP_AUTORESET
            def enqueue(sh):
                for s in range(K):
                    b.step_bbox_ptr(bbox[s].data_ptr(), ops[s].data_ptr(), FL, sh)
            sec, _ = bench.graph_time(dev, enqueue, K, reps=7, warm=4)
            res.append(sec * 1e6)
            del b
            torch.cuda.empty_cache()
        print(f"n={n:7d} K={K:3d}  grouped=0: {res[0]:7.2f} {res[2]:7.2f}   grouped=1: {res[1]:7.2f} {res[3]:7.2f}  us per step", flush=True)
