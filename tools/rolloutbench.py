#!/usr/bin/env python3
"""Throughput of the rollout kernel (T steps per launch, state resident on chip) vs T single-step launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); N = 8192
for T in (16, 64, 256):
    b = EnvBatch(N, 30, 30, -1, "o2arc", dev)
    b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    b.set_tasks_padded(*bench.make_tasks(N, 1)); b.reset()
    bn, on = bench.make_actions(T, N, 7)
    bb, oo = torch.from_numpy(bn).to(dev), torch.from_numpy(on).to(dev)
    FL = (b.elide_flag | 1) if os.environ.get("ROLL_FLAGS", "hot") == "hot" else 0   # (ARCVecEnv's flag set: the lean instantiation)
    b.rollout(bb, oo, FL); torch.cuda.synchronize()
    reps = max(2, 2048 // T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.rollout(bb, oo, FL)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"rollout T={T:4d}: {us:9.1f} us/launch = {us/T:6.2f} us per step-batch -> {N*T/us:8.1f} M env-steps/s", flush=True)
