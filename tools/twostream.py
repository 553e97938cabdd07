#!/usr/bin/env python3
"""Experiment: the 8192-env batch as G independent groups stepped on G HIP streams (launch overhead and the
latency chain of one group overlap the other groups' work).  Prints env-steps/s for G = 1, 2, 4."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); N = 8192; K = 400
for G in (1, 2, 4, 1, 2, 4):
    n = N // G
    streams = [torch.cuda.Stream(dev) for _ in range(G)]
    batches, bb, oo = [], [], []
    for g in range(G):
        b = EnvBatch(n, 30, 30, -1, "o2arc", dev)
        b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
        b.set_tasks_padded(*bench.make_tasks(n, 1 + g)); b.reset()
        bn, on = bench.make_actions(K, n, 7 + g)
        batches.append(b); bb.append(torch.from_numpy(bn).to(dev)); oo.append(torch.from_numpy(on).to(dev))
    torch.cuda.synchronize()
    ptrs = [[(bb[g][i].data_ptr(), oo[g][i].data_ptr()) for i in range(K)] for g in range(G)]
    sh = [s.cuda_stream for s in streams]
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            for g in range(G):
                batches[g].step_bbox_ptr(ptrs[g][i][0], ptrs[g][i][1], 0, sh[g])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"groups={G} envs/group={n}: {N*K/dt/1e6:8.1f} M env-steps/s  ({dt/K*1e6:.2f} us per step of {N} envs)", flush=True)
