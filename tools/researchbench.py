#!/usr/bin/env python3
"""The research-env step (bench.py's research_env leg) alone: us per step of 8192 envs, rows rewritten in full / incrementally.
Used for A/B runs of kernel variants (ARCLE_HIP_LIB=<alternative .so> python tools/researchbench.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); n = 8192; K = 200
bbox_np, op_np = bench.make_actions(K, n, 2000)
bbox, op = torch.from_numpy(bbox_np).to(dev), torch.from_numpy(op_np).to(dev)
r = bench.research_env_leg(dev, n, bbox, op, K)
print(os.environ.get("ARCLE_HIP_LIB", "default"), "full", round(r["rows_rewritten_in_full"]["us_per_step_batch"], 2), "incremental", round(r["rows_incremental"]["us_per_step_batch"], 2),
      "hinted", round(r.get("rows_incremental_hinted_us", 0), 2), "issued MB", round(r["rows_rewritten_in_full"]["roofline"]["traffic"] / 1e6, 1), round(r["rows_incremental"]["roofline"]["traffic"] / 1e6, 1), flush=True)
