#!/usr/bin/env python3
"""Single-env Gym class, step() latency by max_grid_size (30x30: one wavefront per env; beyond 1024 cells: one workgroup per env)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcle_amd.envs import O2ARCv2Env
from arcle_amd.loaders import SyntheticLoader
for size in (30, 40, 64, 127):
    env = O2ARCv2Env(data_loader=SyntheticLoader(n_tasks=4, seed=1, max_size=(size, size)), max_grid_size=(size, size), colors=10)
    env.reset()
    rng = np.random.default_rng(0)
    acts = []
    for _ in range(64):
        sel = np.zeros((size, size), np.int8); x, y = rng.integers(0, size - 4, 2); sel[x:x + 3, y:y + 4] = 1
        acts.append({"selection": sel, "operation": int(rng.integers(0, 34))})
    for a in acts[:16]: env.step(a)
    t = time.perf_counter()
    for k in range(300): env.step(acts[k % 64])
    print(f"max_grid_size {size}x{size}: step() {(time.perf_counter() - t) / 300 * 1e6:.1f} us", flush=True)
