#!/usr/bin/env python3
"""bench.py — env-steps/sec of the ARCLE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
(`python bench.py --gpus N` without a launcher spawns the N ranks itself.)

A "step" is ONE pass of the hot path over one batch: a single `arcle_step_bbox` launch that applies one (selection,
operation) action to every env of this GPU.  Tasks, state and the whole action stream are resident in HBM before the
timed region starts.  Envs are independent, so N GPUs = N shards, no data-path collective (weak scaling); rank 0
prints ONE JSON line.  Workloads (`config.workload` names the one that ran):
  c3 (default, the headline: BASELINE configs[2])  O2ARCv2Env 30x30, 8192 envs/GPU, 35 ops uniform, BBox 5-tuples
      uniform, on-device auto-reset of terminated envs
  c2  O2ARCv2Env 10x10, 1024 envs, ops 0-23, 50 % rectangle / 40 % point / 10 % empty selections
  c4  c3 + the per-step gather a central learner needs: (grid, grid_dim, reward, terminated) packed into ONE
      all_gather_into_tensor over RCCL (BASELINE configs[3])
  c5  ARCEnv 27-op table 30x30, 4096 envs/GPU, 70 % FloodFill point seeds on large-region grids (BASELINE configs[4])

Timing: after an untimed clock ramp and W warm-up steps, the region "barrier + synchronize, exactly K steps, synchronize
+ barrier" is run R times (R reported as `timing.regions`); `value`/`ms_per_step` are the MEDIAN region (max over
ranks per region).  A region that is one hipGraph replay is clocked by HIP events recorded on the launch stream between the two
synchronisations (device time of exactly the K steps; the host-clock figure of the same regions is `timing.host_region_ms`);
eagerly launched regions (multi-rank c4) are clocked by the host.  Besides the contract fields the line carries
  roofline      achieved ALGORITHMIC HBM bytes/s of the step kernel: bytes from the kernel's own per-env accounting
                (planes semantically read+written by the executed op/mode + 56 B) divided by the kernel's
                average launch duration, measured with a HIP-event pair recorded on the launch stream around the K
                back-to-back launches of a timed region; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes
                of this same command (profiles/pmc_latest.json — a recorded figure, labelled as such);
  cpu_baseline  (N=1) on this box's host, bounded samples of the same workload: the oracle's C restatement (1 thread /
                all cores) and `numpy_step` = a plain-NumPy one-env-at-a-time step() loop with the reference's call
                structure (oracle/numpy_env.py), 1 process / all cores.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29e12 measured copy)
STEP_AUTORESET = 1
STEP_PACK_OBS = 256


# ---------------------------------------------------------------------------------------------------------------
# synthetic workloads
# ---------------------------------------------------------------------------------------------------------------
def make_tasks(n, seed, H=30, W=30, lo=1, zero_frac=0.0):
    """ARC-shaped tasks: input dims U{lo..H}x U{lo..W}, colours U{0..9} (a fraction forced to 0); answer == input w.p.
    1/2, else an unrelated grid."""
    rng = np.random.default_rng(seed)
    rows, cols = np.arange(H)[None, :, None], np.arange(W)[None, None, :]

    def grids(dims):
        full = rng.integers(0, 10, (n, H, W)).astype(np.int8)
        if zero_frac:
            full[rng.random((n, H, W)) < zero_frac] = 0
        inside = (rows < dims[:, 0, None, None]) & (cols < dims[:, 1, None, None])
        return np.where(inside, full, 0).astype(np.int8)

    idim = np.stack([rng.integers(lo, H + 1, n), rng.integers(lo, W + 1, n)], 1).astype(np.int8)
    inp = grids(idim)
    same = rng.random(n) < 0.5
    adim = np.where(same[:, None], idim, np.stack([rng.integers(lo, H + 1, n), rng.integers(lo, W + 1, n)], 1)).astype(np.int8)
    ans = np.where(same[:, None, None], inp, grids(adim)).astype(np.int8)
    return inp, idim, ans, adim


def make_actions(steps, n, seed, H=30, W=30, n_ops=35):
    rng = np.random.default_rng(seed)
    bbox = np.stack([rng.integers(0, H, (steps, n)), rng.integers(0, W, (steps, n)),
                     rng.integers(0, H, (steps, n)), rng.integers(0, W, (steps, n))], -1).astype(np.int32)
    op = rng.integers(0, n_ops, (steps, n), dtype=np.int32)  # BBoxWrapper(action_space.sample())
    return bbox, op


def make_actions_c2(steps, n, seed):
    """ops U{0..23}; 50 % rectangle, 40 % point (x1=x2, y1=y2), 10 % empty (corners beyond the plane: the wrapper's
    slices clip to nothing, bbox.py:29)."""
    bbox, op = make_actions(steps, n, seed, 10, 10, 24)
    u = np.random.default_rng(seed + 7).random((steps, n))
    point, empty = (u >= 0.5) & (u < 0.9), u >= 0.9
    bbox[point, 2:] = bbox[point, :2]
    bbox[empty] = 10
    return bbox, op


def _spiral(H, W):
    g = np.full((H, W), 2, np.int8)
    top, left, bot, right = 0, 0, H - 1, W - 1
    while top <= bot and left <= right:  # 1-wide corridor, one cell of field between the arms
        g[top, left:right + 1] = 1
        g[top:bot + 1, right] = 1
        if bot > top + 1:
            g[bot, left + 2:right + 1] = 1
        if right > left + 2 and bot > top + 2:
            g[top + 2:bot + 1, left + 2] = 1
        top, left, bot, right = top + 2, left + 2, bot - 2, right - 2
        if top <= bot and left <= right:
            g[top, left] = 1
    return g


def make_tasks_c5(n, seed, H=30, W=30):
    """Large same-colour regions: stripes, <=3-colour low-frequency blobs, a spiral corridor (long frontier chains)."""
    rng = np.random.default_rng(seed)
    inp = np.zeros((n, H, W), np.int8)
    kind = rng.integers(0, 3, n)
    ii, jj = np.arange(H)[:, None], np.arange(W)[None, :]
    for k in range(n):
        if kind[k] == 0:
            p = int(rng.integers(2, 6))
            inp[k] = (((ii // p) + (jj // p if rng.random() < 0.3 else 0)) % 2) * int(rng.integers(1, 10))
        elif kind[k] == 1:
            coarse = rng.integers(0, 3, (H // 5 + 1, W // 5 + 1))
            inp[k] = np.kron(coarse, np.ones((5, 5), np.int64))[:H, :W] + 1
        else:
            inp[k] = _spiral(H, W)
    dims = np.tile(np.array([[H, W]], np.int8), (n, 1))
    return inp, dims, inp.copy(), dims.copy()


def make_actions_c5(steps, n, seed, H=30, W=30):
    """27-op ARCEnv table: 70 % FloodFill (ops 10-19) from an in-bounds point seed, 30 % any op with a random rectangle."""
    bbox, op = make_actions(steps, n, seed, H, W, 27)
    ff = np.random.default_rng(seed + 3).random((steps, n)) < 0.7
    op[ff] = 10 + (op[ff] % 10)
    bbox[ff, 2:] = bbox[ff, :2]
    return bbox, op


def frontier_rounds_sample(grids, seeds, limit=64):
    """Frontier rounds a flood fill from `seeds` needs (= eccentricity of the seed inside its region), host NumPy, sample."""
    out = []
    for g, (x, y) in list(zip(grids, seeds))[:limit]:
        same = g == g[x, y]
        f = np.zeros_like(same)
        f[x, y] = True
        r = 0
        while True:
            grow = f.copy()
            grow[1:] |= f[:-1]
            grow[:-1] |= f[1:]
            grow[:, 1:] |= f[:, :-1]
            grow[:, :-1] |= f[:, 1:]
            grow &= same
            if (grow == f).all():
                break
            f, r = grow, r + 1
        out.append(r)
    return {"sample": len(out), "mean": float(np.mean(out)), "p90": float(np.percentile(out, 90)), "max": int(np.max(out))}


CONFIGS = {
    "c3": dict(kind="o2arc", H=30, W=30, envs=8192, max_trial=-1, flags=STEP_AUTORESET,
               name="BASELINE configs[2]: O2ARCv2Env 30x30, 8192 envs/GPU, full 35-op O2ARC table uniform, BBoxWrapper "
                    "5-tuples uniform (fused bbox ingress), max_trial=-1, on-device auto-reset of terminated envs"),
    "c2": dict(kind="o2arc", H=10, W=10, envs=1024, max_trial=-1, flags=STEP_AUTORESET,
               name="BASELINE configs[1]: O2ARCv2Env 10x10, 1024 envs, ops 0-23 (Color/FloodFill/Move) uniform, "
                    "50 % rectangle / 40 % point / 10 % empty selections"),
    "c4": dict(kind="o2arc", H=30, W=30, envs=8192, max_trial=-1, flags=STEP_AUTORESET,
               name="BASELINE configs[3]: c3 sharded 8192 envs/GPU + per-step gather of (grid, grid_dim, reward, terminated) "
                    "packed into one RCCL all-gather"),
    "c5": dict(kind="arc", H=30, W=30, envs=4096, max_trial=-1, flags=STEP_AUTORESET,
               name="BASELINE configs[4]: ARCEnv 27-op table 30x30, 4096 envs/GPU, 70 % FloodFill point seeds on "
                    "large-region grids (stripes / 3-colour blobs / spiral), 30 % other ops, max_trial=-1"),
}


# ---------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1): bounded samples, the oracle as the thing TIMED is allowed only here
# ---------------------------------------------------------------------------------------------------------------
def _cpu_run(threads, seed, budget_s):
    from oracle import oracle as O
    O.set_threads(threads)
    n = 2048 if threads == 1 else 8192
    env = O.OracleEnv(n, 30, 30, -1, "o2arc")
    inp, idim, ans, adim = make_tasks(n, seed)
    env.planes["input"][:] = inp
    env.planes["answer"][:] = ans
    env.field("input_dim")[:] = idim
    env.field("answer_dim")[:] = adim
    env.reset()
    chunk = 64
    bbox, op = make_actions(chunk, n, seed + 1)
    done, t0 = 0, time.perf_counter()
    while True:
        for s in range(chunk):
            env.step_bbox(bbox[s], op[s])
        done += chunk * n
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    O.set_threads(1)
    return done / dt, n, done // n


def _numpy_baseline(avail):
    """The plain-NumPy per-env step() loop (oracle/numpy_env.py): one process, then one process per host core."""
    import multiprocessing as mp
    from oracle import numpy_env as NE
    done, sec = NE.run_chunk((11, 64, 150, 30, 30))  # ~10 k steps: a few tenths of a second per 10 k
    steps1 = max(50, int(150 * 6.0 / max(sec, 1e-3)))  # aim at ~6 s of single-process stepping
    done, sec = NE.run_chunk((12, 64, min(steps1, 4000), 30, 30))
    one = done / sec
    procs = max(1, min(avail, 64))
    per = max(20, int(one * 5.0 / 64))  # ~5 s per worker
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(NE.run_chunk, [(100 + i, 64, per, 30, 30) for i in range(procs)])
    wall = time.perf_counter() - t0
    return {"value": one, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"64 envs x {done // 64} C3 steps, one Python step() per env, oracle/numpy_env.py",
            "all_cores": {"value": sum(d for d, _ in res) / max(max(s for _, s in res), 1e-9), "unit": "env-steps/s",
                          "cores": procs, "sample": f"{procs} processes x 64 envs x {per} steps (wall incl. fork {wall:.1f} s)"}}


def cpu_baseline(seed):
    v1, n1, s1 = _cpu_run(1, seed, 10.0)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, 64))
    vt, nt, st = _cpu_run(threads, seed, 6.0)
    return {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n1} envs x {s1} steps of the same C3 action stream, oracle/arcle_oracle.c, 1 thread",
            "host_cores_available": avail,
            "all_cores": {"value": vt, "unit": "env-steps/s", "cores": threads,
                          "sample": f"{nt} envs x {st} steps, same code, OpenMP over envs"},
            "numpy_step": _numpy_baseline(avail),
            "reference_note": "the reference itself cannot travel to this box; survey container: 36 k env-steps/s/core "
                              "(Xeon 2.10 GHz, BASELINE.md §2)"}


def rollout_leg(batch, bbox, op, dev, T=128, reps=8):
    """NOT the headline metric: the same action stream replayed with arcle_rollout_bbox (T steps per launch, env state
    resident in registers between the steps; only per-step reward/terminated and the final state reach HBM)."""
    T = min(T, bbox.shape[0])
    bb, oo = bbox[:T].contiguous(), op[:T].contiguous()
    batch.rollout(bb, oo)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        batch.rollout(bb, oo)
    e1.record()
    torch.cuda.synchronize(dev)
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    return {"mode": "arcle_rollout_bbox", "steps_per_launch": T, "value": T * batch.N / sec, "unit": "env-steps/s",
            "us_per_step_batch": sec / T * 1e6,
            "note": "state stays on chip between steps; not comparable with the per-step HBM roofline above"}


def host_actions_leg(batch, bbox, op, FL, dev, K=200, reps=3):
    """NOT the headline metric: the same steps with the action batch (bbox int32 [N,4] + op int32 [N] = 20 B per env) copied from
    pinned HOST memory before every step, on the launch stream (graph-replayed: 2 copy nodes + 1 kernel node per step) — the
    PCIe-inclusive rate a host-resident policy would see."""
    K = min(K, bbox.shape[0])
    hb, ho = bbox[:K].cpu().pin_memory(), op[:K].cpu().pin_memory()
    db, do = torch.empty_like(bbox[0]), torch.empty_like(op[0])
    st = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        csh = torch.cuda.current_stream(dev).cuda_stream
        for i in range(K):
            db.copy_(hb[i], non_blocking=True)
            do.copy_(ho[i], non_blocking=True)
            batch.step_bbox_ptr(db.data_ptr(), do.data_ptr(), FL, csh)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1) * 1e-3 / K)
    sec = float(np.median(ts))
    return {"mode": "bbox + op copied from pinned host memory before every step (20 B per env)", "value": batch.N / sec,
            "unit": "env-steps/s", "us_per_step_batch": sec * 1e6, "host_bytes_per_step": int(batch.N * 20)}


def research_env_leg(dev, n, bbox, op, K=200, reps=3):
    """NOT the headline metric: the step the reference's training script runs (agents/env.py + agents/train.py:61-68) — op 33 =
    crop, dense reward, TimeLimit(100) truncation, next-step autoreset onto a NEW device-drawn task with colour-permutation +
    rot90 augmentation, and the FilterO2ARC + FlattenObservation row of every env written by the step kernel — all in ONE launch
    per step (feature instantiation), graph-replayed."""
    from arcle_amd import actions
    from arcle_amd.engine import STEP_FLAT_OBS
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader

    class Crop(O2ARCv2Env):  # agents/env.py:23-28
        def create_operations(self):
            ops = super().create_operations()
            ops[33] = actions.reset_sel(actions.crop_grid)
            return ops
    v = ARCVecEnv(Crop, n, SyntheticLoader(n_tasks=400, seed=1, max_size=(30, 30)), device=dev, seed=7, autoreset="resample", augment=("permute", "rot90"),
                  dense_reward=True, max_episode_steps=100)
    v.reset()
    rows = v.batch.set_flat_output(filtered=True)
    FL = v.flags | STEP_FLAT_OBS
    K = min(K, bbox.shape[0])
    st = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        csh = torch.cuda.current_stream(dev).cuda_stream
        for i in range(K):
            v.batch.step_bbox_ptr(bbox[i].data_ptr(), op[i].data_ptr(), FL, csh)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1) * 1e-3 / K)
    sec = float(np.median(ts))
    assert v.batch.status() == 0
    return {"mode": "ARCVecEnv(autoreset='resample', augment, dense_reward, max_episode_steps=100) + fused FilterO2ARC rows",
            "value": n / sec, "unit": "env-steps/s", "us_per_step_batch": sec * 1e6, "row_bytes": int(rows.shape[1]),
            "note": "one launch per step; 2710-byte observation row per env and step written by the step kernel"}


# ---------------------------------------------------------------------------------------------------------------
def _spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per rank)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--envs-per-gpu", type=int, default=0, help="override the config's batch size (kernel sweeps)")
    ap.add_argument("--regions", type=int, default=0, help="timed regions of K steps (default: 5, more for small K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the non-headline legs (kernel A/B runs)")
    ap.add_argument("--no-graph", action="store_true", help="launch the K steps of a region eagerly instead of as one hipGraph")
    ap.add_argument("--no-ramp", action="store_true", help="skip the untimed clock-ramp launches (counter-collection runs)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    dev = torch.device(f"cuda:{local_rank % ndev}")
    torch.cuda.set_device(dev)
    dist = None
    shared_gpu = world > ndev  # more ranks than GPUs (functional test of the N>1 path on a small box): gloo control plane
    if world > 1 or os.environ.get("ARCLE_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv, O2ARCv2Env

    cfg = CONFIGS[a.config]
    H, W, kind = cfg["H"], cfg["W"], cfg["kind"]
    n = a.envs_per_gpu or cfg["envs"]
    K, Wm = a.steps, a.warmup
    R = a.regions or max(5, min(50, math.ceil(2000 / max(K, 1))))
    S = min(K * R + Wm, max(K + Wm, 2048))  # distinct action batches staged in HBM (regions cycle through them)

    # shard = contiguous global env ids [rank*n, (rank+1)*n); per-shard seeds keyed by rank
    batch = EnvBatch(n, H, W, cfg["max_trial"], kind, dev)
    cls = ARCEnv if kind == "arc" else O2ARCv2Env
    batch.set_op_table(actions.table_descs(cls.default_operations()))
    FL = batch.elide_flag | cfg["flags"]  # what ARCVecEnv passes (the state only evolves through the kernels)
    if a.config == "c5":
        tasks = make_tasks_c5(n, 1000 + rank, H, W)
        bbox_np, op_np = make_actions_c5(S, n, 2000 + rank, H, W)
    elif a.config == "c2":
        tasks = make_tasks(n, 1000 + rank, H, W, lo=3, zero_frac=0.5)
        bbox_np, op_np = make_actions_c2(S, n, 2000 + rank)
    else:
        tasks = make_tasks(n, 1000 + rank, H, W)
        bbox_np, op_np = make_actions(S, n, 2000 + rank, H, W)
    batch.set_tasks_padded(*tasks)
    batch.reset()
    bbox = torch.from_numpy(bbox_np).to(dev)
    op = torch.from_numpy(op_np).to(dev)
    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream
    bptr = [bbox[i].data_ptr() for i in range(S)]
    optr = [op[i].data_ptr() for i in range(S)]

    gather = None
    if a.config == "c4":  # the learner-side gather: ONE all-gather of the packed 912-byte record per env and step
        # the step kernel writes the packed rows itself (STEP_PACK_OBS, fused epilogue): no packing launch
        packed = batch.set_packed_output()
        FL |= STEP_PACK_OBS
        full = torch.empty((world * n, packed.shape[1]), dtype=torch.uint8, device=dev)
        host_full = torch.empty(full.shape, dtype=torch.uint8) if shared_gpu else None

        def gather():
            if dist is None:
                return  # one rank: the packed rows ARE the gathered tensor
            if shared_gpu:
                dist.all_gather_into_tensor(host_full, packed.cpu())
            else:
                dist.all_gather_into_tensor(full, packed)

    def step(i):
        j = i % S
        batch.step_bbox_ptr(bptr[j], optr[j], FL, sh)
        if gather is not None:
            gather()

    def barrier():
        if dist is not None:
            dist.barrier()

    def wait_gpu(ev):
        while not ev.query():  # short busy wait: a blocking synchronize wakes up tens of us late
            pass
        torch.cuda.synchronize(dev)

    # ---- untimed: W warm-up steps ---------------------------------------------------------------------------------
    for i in range(Wm):
        step(i)
    torch.cuda.synchronize(dev)
    # snapshot of the state region 0 starts from (replayed below for the byte accounting)
    snap = {k: v.clone() for k, v in batch.planes.items()}
    snap_rec, snap_cnt = batch.rec.clone(), batch.cnt.clone()

    # The K steps of a region are captured once into a hipGraph (K launches of arcle_step_kernel, each with its own
    # action batch) and replayed per region: a launch-bound inner loop belongs in a graph, and the host then issues one
    # call per region instead of K.  (With more than one rank c4 keeps eager launches: its collective is not captured.)
    graph = None
    if not a.no_graph and (gather is None or dist is None):
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=torch.cuda.Stream(dev)):
                csh = torch.cuda.current_stream(dev).cuda_stream
                for i in range(Wm, Wm + K):
                    batch.step_bbox_ptr(bptr[i % S], optr[i % S], FL, csh)
        except Exception as exc:  # capture unsupported: eager launches
            print(f"bench: hipGraph capture failed ({exc}); eager launches", file=sys.stderr)
            graph = None

    def region(r):
        if graph is not None:
            graph.replay()
        else:
            for i in range(Wm + r * K, Wm + (r + 1) * K):
                step(i)

    # untimed clock ramp (the chip idles at low clocks before the first launch): the region's own launches, ~80 ms of them
    t_ramp, r = time.perf_counter(), 0
    n_ramp = max(2, min(50, 4000 // max(K, 1)))  # (a fixed count when ranks must stay in step with each other)
    while not a.no_ramp and ((time.perf_counter() - t_ramp < 0.08) if dist is None else (r < n_ramp)):
        region(r)
        r += 1
        torch.cuda.synchronize(dev)
    for k, v in snap.items():  # back to the state the regions are defined to start from
        batch.planes[k].copy_(v)
    batch.rec.copy_(snap_rec)
    batch.cnt.copy_(snap_cnt)
    torch.cuda.synchronize(dev)

    # ---- R timed regions of exactly K steps, each bracketed by barrier + synchronize ---------------------------
    # Clock of a region: the HIP events recorded on the launch stream right inside the two synchronisations when the region
    # is ONE hipGraph replay (device time of exactly the K steps), the host clock otherwise.  The host figure additionally
    # holds one graph-launch submission and one wake-up (~20 us per region, i.e. 17 % of a 20-step region and 1 % of a
    # 400-step one) — it is reported beside it (timing.host_region_ms), never instead of work.
    device_clock = graph is not None
    wall, kern, devt = [], [], []
    for r in range(R):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev0.record(stream)  # HIP events on the stream the kernel is launched on
        region(r)
        ev1.record(stream)
        wait_gpu(ev1)
        barrier()
        wall.append(time.perf_counter() - t0)
        devt.append(ev0.elapsed_time(ev1) * 1e-3)
        kern.append(devt[-1] / K)
    host_ms = [round(x * 1e3, 4) for x in wall][:12]
    wall_t = torch.tensor(devt if device_clock else wall, dtype=torch.float64)
    if dist is not None:  # max over ranks, per region
        wt = wall_t if shared_gpu else wall_t.to(dev)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        wall_t = wt.cpu()
    elapsed = float(wall_t.median())
    kernel_avg_s = float(np.median(kern))
    status = batch.status()
    assert status == 0, f"device status {status}"
    total_steps = K * n * world

    # ---- algorithmic bytes of the K launches of region 0: restore the snapshot and replay them (untimed) with the
    #      kernel's per-env byte accounting switched on ----------------------------------------------------------
    roofline = None
    if rank == 0:
        for k, v in snap.items():
            batch.planes[k].copy_(v)
        batch.rec.copy_(snap_rec)
        batch.cnt.copy_(snap_cnt)
        batch.enable_accounting(True)
        batch.accounting(clear=True)
        for i in range(Wm, Wm + K):
            j = i % S
            batch.step_bbox_ptr(bptr[j], optr[j], FL & ~STEP_PACK_OBS, sh)  # (the accounting instantiation has no epilogues)
        torch.cuda.synchronize(dev)
        nbytes, nsteps = batch.accounting(clear=True)
        batch.enable_accounting(False)
        if FL & STEP_PACK_OBS:  # the packed row: the grid plane read once more, the row written once
            nbytes += K * n * (H * W + batch.packed_obs_size())
        per_launch_bytes = nbytes / K
        achieved = per_launch_bytes / kernel_avg_s
        traffic = traffic_src = frac_traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and a.config == "c3" and n == CONFIGS["c3"]["envs"]:
            pmc = json.load(open(pmc_path))
            traffic = pmc.get("hbm_bytes_per_launch")
            traffic_src = "recorded (not measured in this run): " + str(pmc.get("source"))
            if traffic:
                frac_traffic = traffic / kernel_avg_s / HBM_PEAK
        roofline = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                    "frac_by_traffic": frac_traffic,
                    "kernel": "arcle_step_kernel" + (" (feature instantiation with the fused packed-row epilogue; the all-gather is inside the event pair)"
                                                     if gather is not None else ""),
                    "avg_launch_us": kernel_avg_s * 1e6,
                    "algorithmic_bytes_per_launch": per_launch_bytes,
                    "algorithmic_bytes_per_env_step": nbytes / max(nsteps, 1),
                    "note": "algorithmic bytes follow SURVEY.md 8d and include the reset_sel zero-fills of `selected` that "
                            "ARCLE_STEP_ELIDE_SELECTED never writes (about 14 % of the figure on this mix)",
                    "frac_of_measured_copy_peak_6.29TBps": achieved / 6.29e12}

    if rank == 0:
        out = {
            "metric": "env-steps/sec (whole node), O2ARCv2Env 30x30, 8192 envs/GPU" if a.config in ("c3", "c4")
                      else f"env-steps/sec (whole node), config {a.config}",
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic",
            "config": {"workload": cfg["name"], "id": a.config, "envs_per_gpu": n, "global_envs": n * world,
                       "grid": [H, W], "ingress": "bbox",
                       "parallelism": f"env-shard x{world} (no data-path collective)" if gather is None
                       else (f"env-shard x{world} + one packed all-gather per step ({'gloo, shared GPU' if shared_gpu else 'RCCL'})"
                             if dist is not None else "one rank: step with the fused packed-row epilogue, nothing to gather")},
            "timing": {"regions": R, "stat": "median region, max over ranks per region",
                       "clock": "HIP events on the launch stream, recorded between the region's two synchronisations" if device_clock else "host perf_counter between the region's two synchronisations",
                       "host_region_ms": host_ms,
                       "launch": "hipGraph of the K step launches, one replay per region" if graph is not None else "eager",
                       "region_ms": [round(float(x) * 1e3, 4) for x in wall_t.tolist()][:12]},
            "roofline": roofline,
        }
        if a.config == "c5":
            seeds = [(int(bbox_np[0, e, 0]), int(bbox_np[0, e, 1])) for e in range(n) if 10 <= op_np[0, e] < 20]
            grids = [tasks[0][e] for e in range(n) if 10 <= op_np[0, e] < 20]
            out["floodfill"] = {"share_of_actions": float(((op_np >= 10) & (op_np < 20)).mean()),
                                "frontier_rounds": frontier_rounds_sample(grids, seeds)}
        if world == 1 and not a.no_extras and a.config == "c3":
            out["extras"] = {"rollout": rollout_leg(batch, bbox, op, dev), "research_env": research_env_leg(dev, n, bbox, op),
                             "host_actions": host_actions_leg(batch, bbox, op, FL, dev)}
        if world == 1 and not a.no_cpu_baseline and a.config == "c3":
            out["cpu_baseline"] = cpu_baseline(1000)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
