#!/usr/bin/env python3
"""Host-side cost of the Python front-end: ARCVecEnv.step_bbox called from a plain Python loop (no graph), us per call at 8192 envs,
against the raw ABI call (EnvBatch.step_bbox_ptr) in the same loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
from arcle_amd.loaders import SyntheticLoader
n, K = 8192, 300
v = ARCVecEnv(O2ARCv2Env, n, SyntheticLoader(n_tasks=200, seed=1), autoreset=True, seed=3)
v.reset()
bbox_np, op_np = bench.make_actions(K, n, 5)
bbox, op = torch.from_numpy(bbox_np).cuda(), torch.from_numpy(op_np).cuda()
for name, fn in (("ARCVecEnv.step_bbox", lambda i: v.step_bbox(bbox[i], op[i])),
                 ("EnvBatch.step_bbox", lambda i: v.batch.step_bbox(bbox[i], op[i], v.flags)),
                 ("EnvBatch.step_bbox_ptr", lambda i: v.batch.step_bbox_ptr(bbox[i].data_ptr(), op[i].data_ptr(), v.flags, 0))):
    for i in range(20): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): fn(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:26s} {dt / K * 1e6:7.2f} us per call  ({n * K / dt / 1e6:7.1f} M env-steps/s)", flush=True)
