#!/usr/bin/env python3
"""bench.py's host_actions leg alone (A/B of library builds through ARCLE_HIP_LIB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); n = 8192; K = 200
bb, op = bench.make_actions(K, n, 2000)
r = bench.host_actions_leg(dev, n, torch.from_numpy(bb).to(dev), torch.from_numpy(op).to(dev))
print(os.environ.get("ARCLE_HIP_LIB", "default"), {k: round(v["us_per_step_batch"], 2) for k, v in r.items() if isinstance(v, dict) and "us_per_step_batch" in v}, flush=True)
