#!/usr/bin/env python3
"""bench.py's transition_rows leg alone, out of place and in place (A/B of library builds through ARCLE_HIP_LIB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); n = 8192
bb, op = bench.make_actions(64, n, 2000)
bbox, ops = torch.from_numpy(bb).to(dev), torch.from_numpy(op).to(dev)
r = bench.transition_leg(dev, n, bbox, ops)
print(os.environ.get("ARCLE_HIP_LIB", "default"), "out of place", round(r["us_per_step_batch"], 2), "us  frac", round(r["roofline"]["frac"], 3),
      " densely packed rows", round(r["rows_densely_packed"]["us_per_step_batch"], 2), "us  in place (leg)", round(r["in_place"]["us_per_step_batch"], 2), flush=True)
# in place: rows_out is rows_in
batch = bench.make_batch(dev, n)
rows = batch.get_state_rows().clone()
buf = torch.zeros((n, (rows.shape[1] + 15) & ~15), dtype=torch.int8, device=dev); buf[:, :rows.shape[1]] = rows
rw, tm = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
K = 32
def enqueue(sh):
    for i in range(K):
        batch._check(batch.L.arcle_transition_rows(batch._h, n, buf.data_ptr(), buf.shape[1], 1, bbox[i].data_ptr(), ops[i].data_ptr(), None,
                                                   buf.data_ptr(), buf.shape[1], 0, rw.data_ptr(), tm.data_ptr(), 0, sh), "arcle_transition_rows")
sec, _ = bench.graph_time(dev, enqueue, K)
print("in place", round(sec * 1e6, 2), "us", flush=True)
