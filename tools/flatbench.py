#!/usr/bin/env python3
"""Cost of the observation rows per step at 8192 envs (C3 actions, graph-replayed): step alone, step with the fused writer
(STEP_FLAT_OBS), step followed by the stand-alone flatten launch — full (6314 B) and FilterO2ARC (2710 B) rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch, STEP_FLAT_OBS
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = 8192; K = 200
bbox_np, op_np = bench.make_actions(K, n, 5)
bbox = torch.from_numpy(bbox_np).to(dev); ops = torch.from_numpy(op_np).to(dev)
for filtered in (False, True):
    for mode in ("step only", "fused", "step + flatten launch"):
        batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
        batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
        batch.set_tasks_padded(*bench.make_tasks(n, 1)); batch.reset()
        batch.set_flat_output(filtered)
        FL = batch.elide_flag | bench.STEP_AUTORESET
        st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            sh = torch.cuda.current_stream(dev).cuda_stream
            for i in range(K):
                batch.step_bbox_ptr(bbox[i].data_ptr(), ops[i].data_ptr(), FL | (STEP_FLAT_OBS if mode == "fused" else 0), sh)
                if mode.endswith("launch"):
                    batch.flat_obs(out=batch._flat_buf, filtered=filtered)
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / K * 1e3)
        print(f"rows {'FilterO2ARC 2710 B' if filtered else 'full 6314 B':18s} {mode:24s} {sorted(ts)[2]:6.2f} us per step", flush=True)
