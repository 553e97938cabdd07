#!/usr/bin/env python3
"""The single-env drop-in class (arcle_amd.envs.O2ARCv2Env, one env on the GPU, numpy state dict back on the host after every
step): steps per second from a plain Python loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arcle_amd.envs import O2ARCv2Env
from arcle_amd.loaders import SyntheticLoader
env = O2ARCv2Env(data_loader=SyntheticLoader(n_tasks=20, seed=1), max_grid_size=(30, 30))
obs, info = env.reset()
rng = np.random.default_rng(0)
acts = []
for _ in range(300):
    sel = np.zeros((30, 30), np.int8)
    x, y = rng.integers(0, 25, 2)
    sel[x:x + rng.integers(1, 5), y:y + rng.integers(1, 5)] = 1
    acts.append({"selection": sel, "operation": int(rng.integers(0, 34))})
for a in acts[:20]:
    env.step(a)
t0 = time.perf_counter()
for a in acts:
    obs, r, term, trunc, info = env.step(a)
dt = time.perf_counter() - t0
print(f"O2ARCv2Env.step (one env, state dict on the host): {dt / len(acts) * 1e6:.1f} us per step = {len(acts) / dt:.0f} steps/s")
import copy
st = copy.deepcopy(obs)
for a in acts[:20]:
    env.transition(st, a)
t0 = time.perf_counter()
for a in acts[:200]:
    env.transition(st, a)
print(f"O2ARCv2Env.transition(state, action): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
