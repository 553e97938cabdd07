#!/usr/bin/env python3
"""How much of a launch is the tail of the object ops?  The C3 action stream with (a) nothing changed, (b) Rotate replaced by Flip,
(c) Rotate/Flip replaced by Move, (d) Rotate/Flip/Move replaced by Color — graph-replayed, us per launch of 8192 envs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = 8192; K = 400
bbox_np, op_np = bench.make_actions(K, n, 5)
def variant(v):
    op = op_np.copy()
    if v >= 1: op[(op == 24) | (op == 25)] = 26 + (op[(op == 24) | (op == 25)] & 1)
    if v >= 2: m = (op >= 24) & (op <= 27); op[m] = 20 + (op[m] & 3)
    if v >= 3: m = (op >= 20) & (op <= 27); op[m] = op[m] % 10
    return op
for rep in range(2):
    for v, name in enumerate(["C3 mix", "Rotate->Flip", "Rotate/Flip->Move", "Rotate/Flip/Move->Color"]):
        batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
        batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
        batch.set_tasks_padded(*bench.make_tasks(n, 1)); batch.reset()
        FL = batch.elide_flag | bench.STEP_AUTORESET
        bbox = torch.from_numpy(bbox_np).to(dev); ops = torch.from_numpy(variant(v)).to(dev)
        st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(K):
                batch.step_bbox_ptr(bbox[i].data_ptr(), ops[i].data_ptr(), FL, torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / K * 1e3)
        print(f"{name:26s} {sorted(ts)[3]:.2f} us per launch", flush=True)
