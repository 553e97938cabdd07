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

G, H, W, S = 13, 10, 10, 24  # 13 envs over 2 ranks: shards of 7 and 6 (the short one is padded for the collective)


class _PackedRows:
    """The slice of the EnvBatch interface ShardedVecEnv.gather uses (packed_obs_size / packed_obs / device / H / W), over
    the oracle's arrays; the row layout is arcle_pack_obs's (include/arcle_hip.h)."""
    device = torch.device("cpu")

    def __init__(self, env):
        self.env, self.H, self.W = env, H, W

    def packed_obs_size(self):
        return (H * W + 7 + 15) & ~15

    def packed_obs(self, out):
        if out.shape[0] == 0:
            return out
        be, n = self.env.be, self.env.N
        rows = np.zeros((n, self.packed_obs_size()), np.uint8)
        rows[:, :H * W] = be.get("grid").reshape(n, -1).view(np.uint8)
        rows[:, H * W:H * W + 2] = be.get("grid_dim").view(np.uint8)
        rows[:, H * W + 2:H * W + 6] = self.env.last[0].astype("<i4").view(np.uint8).reshape(n, 4)
        rows[:, H * W + 6] = self.env.last[1]
        out.copy_(torch.from_numpy(rows))
        return out


class OracleVecEnv:
    """ARCVecEnv-shaped adapter over the oracle for global env ids [lo, hi)."""

    def __init__(self, n, lo, hi, tasks):
        self.N = n
        self.lo, self.hi = lo, hi
        self.be = B.OracleBackend(n, H, W, 3, "o2arc", O.o2arc_ops())
        inp, idim, ans, adim = tasks
        self.be.set_tasks(inp[lo:hi], idim[lo:hi], ans[lo:hi], adim[lo:hi])
        self.be.reset()
        self.batch = _PackedRows(self)

    def _obs(self):
        return {"grid": torch.from_numpy(self.be.get("grid")), "grid_dim": torch.from_numpy(self.be.get("grid_dim"))}

    def step_bbox(self, bbox, op):
        r, t = self.be.step("bbox", bbox.numpy(), op.numpy())
        self.last = (r, t)
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
    out = {}
    # (1) synchronous gather after every step, unequal shards
    env = ShardedVecEnv(G, lambda n, lo, hi: OracleVecEnv(n, lo, hi, tasks))
    assert (env.lo, env.hi) == shard_range(G, world, rank) and env.N == env.hi - env.lo
    grids, rewards = [], []
    for s in range(S):
        obs, r, t, _, _ = env.step_bbox(env.local_slice(bbox[s]), env.local_slice(op[s]))  # local, no comm
        ggrid, gdim, gr, gt = env.gather()                                               # the only collective: ONE all-gather
        assert ggrid.shape == (G, H, W) and gdim.shape == (G, 2) and gr.shape == (G,) and gt.dtype == torch.bool
        grids.append(ggrid.numpy().copy())
        rewards.append(gr.numpy().copy())
    out["sync"] = (np.stack(grids), np.stack(rewards))
    # (2) overlapped: the gather of step s is only waited for AFTER step s+1 has been issued (double-buffered rows)
    env = ShardedVecEnv(G, lambda n, lo, hi: OracleVecEnv(n, lo, hi, tasks))
    grids, pending = [], None
    for s in range(S):
        env.step_bbox(env.local_slice(bbox[s]), env.local_slice(op[s]))
        work = env.gather_async()
        if pending is not None:
            grids.append(pending.wait()[0].numpy().copy())
            env.release(pending)
        pending = work
    grids.append(pending.wait()[0].numpy().copy())
    out["async"] = np.stack(grids)
    # (3) two ping-pong groups: group B steps while group A's rows are in flight
    env = ShardedVecEnv(G, lambda n, lo, hi: OracleVecEnv(n, lo, hi, tasks), groups=2)
    per_group = [[], []]
    for s in range(S):
        works = []
        for g in range(2):
            env.step_bbox(env.local_slice(bbox[s], g), env.local_slice(op[s], g), group=g)
            works.append(env.gather_async(g))
        for g in range(2):
            per_group[g].append(works[g].wait()[0].numpy().copy())
    out["groups"] = ([np.stack(x) for x in per_group], [env.group_global_ids(g).numpy() for g in range(2)])
    # (4) every = 4: rows of 4 steps travel in one collective
    env = ShardedVecEnv(G, lambda n, lo, hi: OracleVecEnv(n, lo, hi, tasks), every=4)
    chunks = []
    for s in range(S):
        env.step_bbox(env.local_slice(bbox[s]), env.local_slice(op[s]))
        assert env.ready() == (s % 4 == 3)
        if env.ready():
            ggrid, gdim, gr, gt = env.gather()
            assert ggrid.shape == (4, G, H, W) and gr.shape == (4, G)
            chunks.append(ggrid.numpy().copy())
    out["every"] = np.concatenate(chunks)
    # (5) a consumer that never waits for half of its gathers: the work of step s is dropped for even s, and the work of an odd step
    # is only waited for after TWO more steps were issued.  The step that re-enters a slot must itself wait for the collective that
    # still reads that slot's packed rows (ShardedVecEnv._before_step) — the rows SENT are then never rewritten under a gather.
    env = ShardedVecEnv(G, lambda n, lo, hi: OracleVecEnv(n, lo, hi, tasks))
    waited, sparse, held = [], [], None
    grp = env.groups[0]
    orig_before = env._before_step

    def spy(g):
        pend = g.inflight[g.slot] if g.t == 0 else None
        orig_before(g)
        if pend is not None:
            assert g.inflight[g.slot] is None and pend.is_completed(), "the slot's gather must have finished before the slot is rewritten"
            waited.append(1)
    env._before_step = spy
    for s in range(S):
        env.step_bbox(env.local_slice(bbox[s]), env.local_slice(op[s]))
        work = env.gather_async()
        if s % 2 == 1:
            if held is not None:  # the gather of step s-2, two steps (and one slot reuse) late: its OUTPUT buffer has been reused by
                held[1].wait()    # gather s (documented: valid until the gather after the next one) — only completion is checked
            held = (s, work)
            sparse.append(work.wait()[0].numpy().copy())
    out["sparse"] = (np.stack(sparse), len(waited))
    if rank == 0:
        out_q.put(out)
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
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tasks, bbox, op = make_tasks_and_actions()
    single = OracleVecEnv(G, 0, G, tasks)
    want = []
    for s in range(S):
        obs, r, t, _, _ = single.step_bbox(bbox[s], op[s])
        want.append(obs["grid"].numpy().copy())
        assert np.array_equal(want[-1], out["sync"][0][s]), f"gathered grids differ from the 1-process run at step {s}"
        assert np.array_equal(r.numpy(), out["sync"][1][s])
    want = np.stack(want)
    assert np.array_equal(out["async"], want), "overlapped gathers (waited one step late) differ from the 1-process run"
    (ga, gb), (ia, ib) = out["groups"]
    assert sorted(ia.tolist() + ib.tolist()) == list(range(G))
    assert np.array_equal(ga, want[:, ia]) and np.array_equal(gb, want[:, ib]), "ping-pong groups differ from the 1-process run"
    assert np.array_equal(out["every"], want), "every=4 gathers differ from the 1-process run"
    sparse, n_waited = out["sparse"]
    assert np.array_equal(sparse, want[1::2]), "gathers of a consumer that drops every other work differ from the 1-process run"
    assert n_waited >= S - 2, "every step that re-entered a slot must have waited for the gather still reading it"


def test_shard_ranges():
    from arcle_amd.dist import shard_range
    for g, w in ((65536, 8), (8192, 1), (10, 4), (7, 8)):
        spans = [shard_range(g, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == g
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    assert shard_range(65536, 8, 3) == (24576, 32768)
    # (that the device-side task draws follow the GLOBAL env id is tested with the kernels: tests/features.py sampler)
