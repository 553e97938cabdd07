#!/usr/bin/env python3
"""The C3 action stream through the ingress forms (bbox tuples, point tuples, full H x W int8 masks built from the same
rectangles, the same masks bit-packed, 5-tuple records, 5-tuple records read straight from pinned host memory), graph-replayed:
us per launch of 8192 envs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = 8192; K = 64
bbox_np, op_np = bench.make_actions(K, n, 5)
x1 = np.minimum(bbox_np[..., 0], bbox_np[..., 2]); x2 = np.maximum(bbox_np[..., 0], bbox_np[..., 2])
y1 = np.minimum(bbox_np[..., 1], bbox_np[..., 3]); y2 = np.maximum(bbox_np[..., 1], bbox_np[..., 3])
ii = np.arange(30)[None, None, :, None]; jj = np.arange(30)[None, None, None, :]
masks = ((ii >= x1[..., None, None]) & (ii <= x2[..., None, None]) & (jj >= y1[..., None, None]) & (jj <= y2[..., None, None])).astype(np.int8)
pay = {"bbox": torch.from_numpy(bbox_np).to(dev), "point": torch.from_numpy(np.ascontiguousarray(bbox_np[..., :2])).to(dev),
       "mask": torch.from_numpy(masks).to(dev)}
ops = torch.from_numpy(op_np).to(dev)
act5 = np.concatenate([bbox_np, op_np[..., None]], -1).astype(np.int32)
pay["bbox5"] = torch.from_numpy(act5).to(dev)
pay["bbox5_host"] = torch.from_numpy(act5).pin_memory()
for ing in ("bbox", "point", "mask", "bits", "bbox5", "bbox5_host"):
    batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
    batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    batch.set_tasks_padded(*bench.make_tasks(n, 1)); batch.reset()
    FL = batch.elide_flag | bench.STEP_AUTORESET
    if ing == "bits":
        pay["bits"] = torch.stack([batch.pack_mask_bits(pay["mask"][i]) for i in range(K)])
        torch.cuda.synchronize()
    fn = {"bbox": batch.L.arcle_step_bbox, "point": batch.L.arcle_step_point, "mask": batch.L.arcle_step_mask,
          "bits": batch.L.arcle_step_bits}.get(ing)
    st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        sh = torch.cuda.current_stream(dev).cuda_stream
        for i in range(K):
            if fn is None:
                rc = batch.L.arcle_step_bbox5(batch._h, pay[ing][i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh)
            else:
                rc = fn(batch._h, pay[ing][i].data_ptr(), ops[i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh)
            assert rc == 0
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / K * 1e3)
    print(f"{ing:6s} ingress: {sorted(ts)[3]:.2f} us per launch of {n} envs", flush=True)
