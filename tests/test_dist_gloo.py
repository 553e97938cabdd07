"""The N>1 path on CPU: 2 processes over gloo.  The sharding / gather logic of arcle_amd.dist is exercised with
a local vector env backed by the ORACLE (tests only — the product's local env is the HIP ARCVecEnv, which needs
a GPU); the check is that 2 shards + gather reproduce a single-process run of all envs, in global env order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import backends as B
from oracle import oracle as O

G, H, W, S = 12, 10, 10, 24


class OracleVecEnv:
    """ARCVecEnv-shaped adapter over the oracle for global env ids [lo, hi)."""

    def __init__(self, n, lo, hi, tasks):
        self.N = n
        self.be = B.OracleBackend(n, H, W, 3, "o2arc", O.o2arc_ops())
        inp, idim, ans, adim = tasks
        self.be.set_tasks(inp[lo:hi], idim[lo:hi], ans[lo:hi], adim[lo:hi])
        self.be.reset()

    def _obs(self):
        return {"grid": torch.from_numpy(self.be.get("grid")), "grid_dim": torch.from_numpy(self.be.get("grid_dim"))}

    def step_bbox(self, bbox, op):
        r, t = self.be.step("bbox", bbox.numpy(), op.numpy())
        return self._obs(), torch.from_numpy(r), torch.from_numpy(t).bool(), torch.zeros(self.N, dtype=torch.bool), {}


def make_tasks_and_actions():
    rng = np.random.default_rng(42)
    inp = np.zeros((G, H, W), np.int8)
    idim = rng.integers(1, H + 1, (G, 2)).astype(np.int8)
    for n in range(G):
        inp[n, :idim[n, 0], :idim[n, 1]] = rng.integers(0, 10, (idim[n, 0], idim[n, 1]))
    bbox = torch.from_numpy(rng.integers(0, H, (S, G, 4)).astype(np.int32))
    op = torch.from_numpy(rng.integers(0, 35, (S, G)).astype(np.int32))
    return (inp, idim, inp.copy(), idim.copy()), bbox, op


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arcle_amd.dist import ShardedVecEnv, shard_range
    tasks, bbox, op = make_tasks_and_actions()
    env = ShardedVecEnv(G, lambda n, lo, hi: OracleVecEnv(n, lo, hi, tasks))
    assert (env.lo, env.hi) == shard_range(G, world, rank) and env.N == G // world
    grids, rewards = [], []
    for s in range(S):
        obs, r, t, _, _ = env.step_bbox(env.local_slice(bbox[s]), env.local_slice(op[s]))  # local, no comm
        gobs, gr, gt = env.gather(obs, r, t)                                              # the only collective
        assert gobs["grid"].shape == (G, H, W) and gr.shape == (G,) and gt.dtype == torch.bool
        grids.append(gobs["grid"].numpy().copy())
        rewards.append(gr.numpy().copy())
    if rank == 0:
        out_q.put((np.stack(grids), np.stack(rewards)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_shards_equal_one_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    grids, rewards = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tasks, bbox, op = make_tasks_and_actions()
    single = OracleVecEnv(G, 0, G, tasks)
    for s in range(S):
        obs, r, t, _, _ = single.step_bbox(bbox[s], op[s])
        assert np.array_equal(obs["grid"].numpy(), grids[s]), f"gathered grids differ from the 1-process run at step {s}"
        assert np.array_equal(r.numpy(), rewards[s])


def test_shard_ranges_and_seeds():
    from arcle_amd.dist import shard_range, shard_seed
    for g, w in ((65536, 8), (8192, 1), (10, 4), (7, 8)):
        spans = [shard_range(g, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == g
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert shard_range(65536, 8, 3) == (24576, 32768)
    # per-env streams depend on the global id only, not on the sharding
    assert shard_seed(7, 12345) == shard_seed(7, 12345) and shard_seed(7, 1) != shard_seed(7, 2)
