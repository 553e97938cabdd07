#!/usr/bin/env python3
"""Experiment: the env batch as G independent groups, each with its own handle, HIP stream and hipGraph of K steps; the G graphs are
replayed concurrently (one group's launch ramp and tail overlap the other groups' work — envs are independent, so the only ordering
that matters is step k -> step k+1 of the SAME group).  MODE=fork captures all groups into ONE graph with a fork/join instead.
Prints env-steps/s for G = 1, 2, 4, 8 at a fixed total batch N."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); N = int(os.environ.get("N", 8192)); K = 400; MODE = os.environ.get("MODE", "graphs")
for G in (1, 2, 4, 8, 1, 2, 4, 8):
    n = N // G
    side = [torch.cuda.Stream(dev) for _ in range(G)]
    batches, bb, oo = [], [], []
    for g in range(G):
        b = EnvBatch(n, 30, 30, -1, "o2arc", dev)
        b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
        b.set_tasks_padded(*bench.make_tasks(n, 1 + g)); b.reset()
        bn, on = bench.make_actions(K, n, 7 + g)
        batches.append(b); bb.append(torch.from_numpy(bn).to(dev)); oo.append(torch.from_numpy(on).to(dev))
    FL = batches[0].elide_flag | bench.STEP_AUTORESET
    torch.cuda.synchronize()
    graphs = []
    if MODE == "fork":
        graph = torch.cuda.CUDAGraph(); main = torch.cuda.Stream(dev)
        with torch.cuda.graph(graph, stream=main):
            fork = torch.cuda.Event(); fork.record(main)
            for g in range(G):
                side[g].wait_event(fork)
                for i in range(K):
                    batches[g].step_bbox_ptr(bb[g][i].data_ptr(), oo[g][i].data_ptr(), FL, side[g].cuda_stream)
                e = torch.cuda.Event(); e.record(side[g]); main.wait_event(e)
        graphs = [(graph, main)]
    else:
        for g in range(G):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=side[g]):
                for i in range(K):
                    batches[g].step_bbox_ptr(bb[g][i].data_ptr(), oo[g][i].data_ptr(), FL, side[g].cuda_stream)
            graphs.append((gr, side[g]))

    def replay():
        for gr, st in graphs:
            with torch.cuda.stream(st):
                gr.replay()
    for _ in range(30): replay()
    torch.cuda.synchronize()
    ts = []
    for rep in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        replay()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    print(f"{MODE}: groups={G} envs/group={n}: {N*K/dt/1e6:8.1f} M env-steps/s  ({dt/K*1e6:.2f} us per step of {N} envs)", flush=True)
