#!/usr/bin/env python3
"""bench.py — env-steps/sec of the ARCLE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over one batch: a single `arcle_step_bbox` launch that applies one
(bbox, operation) action to each of the 8192 envs of this GPU (BASELINE config 3: O2ARCv2Env 30x30, all 35
ops uniform, BBoxWrapper 5-tuples uniform; synthetic tasks).  Tasks, state and the whole action stream are
resident in HBM before the timed region starts.  Envs are independent, so N GPUs = N shards of 8192 envs,
no data-path collective (weak scaling); rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      achieved algorithmic HBM bytes/s of the step kernel: bytes from the kernel's own per-env
                accounting (planes semantically read+written by the executed op/mode + 56 B)
                divided by the kernel's average launch duration, measured with ONE HIP-event pair recorded on the
                launch stream around the K back-to-back launches of the timed region;
  cpu_baseline  the oracle's C restatement (oracle/arcle_oracle.c, one thread) timed on this box's host on a
                bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 8192
H = W = 30
HBM_PEAK = 8.0e12  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29e12 measured copy)


def make_tasks(n, seed):
    """Synthetic ARC-shaped tasks: input dims U{1..30}^2, colours U{0..9}; answer == input w.p. 1/2, else an
    unrelated grid with dims U{1..30}^2."""
    rng = np.random.default_rng(seed)
    rows, cols = np.arange(H)[None, :, None], np.arange(W)[None, None, :]

    def grids(dims):
        full = rng.integers(0, 10, (n, H, W)).astype(np.int8)
        inside = (rows < dims[:, 0, None, None]) & (cols < dims[:, 1, None, None])
        return np.where(inside, full, 0).astype(np.int8)

    idim = rng.integers(1, 31, (n, 2)).astype(np.int8)
    inp = grids(idim)
    same = rng.random(n) < 0.5
    adim = np.where(same[:, None], idim, rng.integers(1, 31, (n, 2))).astype(np.int8)
    ans = np.where(same[:, None, None], inp, grids(adim)).astype(np.int8)
    return inp, idim, ans, adim


def make_actions(steps, n, seed):
    rng = np.random.default_rng(seed)
    bbox = rng.integers(0, 30, (steps, n, 4), dtype=np.int32)  # BBoxWrapper(action_space.sample())
    op = rng.integers(0, 35, (steps, n), dtype=np.int32)
    return bbox, op


def _cpu_run(threads, seed, budget_s):
    from oracle import oracle as O
    O.set_threads(threads)
    n = 2048 if threads == 1 else 8192
    env = O.OracleEnv(n, H, W, -1, "o2arc")
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


def cpu_baseline(seed):
    """The oracle's C restatement (oracle/arcle_oracle.c) on this box's host cores, same C3 workload, bounded sample:
    `value` is ONE thread; `all_cores` is the same code with its env loop split over host threads (OpenMP)."""
    v1, n1, s1 = _cpu_run(1, seed, 12.0)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(avail, 64))
    vt, nt, st = _cpu_run(threads, seed, 8.0)
    return {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n1} envs x {s1} steps of the same C3 action stream, oracle/arcle_oracle.c, 1 thread",
            "host_cores_available": avail,
            "all_cores": {"value": vt, "unit": "env-steps/s", "cores": threads,
                          "sample": f"{nt} envs x {st} steps, same code, OpenMP over envs"}}


def rollout_leg(batch, bbox, op, start, dev, T=128, reps=8):
    """NOT the headline metric: the same action stream replayed with arcle_rollout_bbox (T steps per launch, env
    state resident in registers between the steps; only per-step reward/terminated and the final state reach HBM).
    For callers that hold the action sequence up front (trace replay, scripted policies)."""
    T = min(T, bbox.shape[0] - start)
    bb, oo = bbox[start:start + T].contiguous(), op[start:start + T].contiguous()
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the non-headline legs (kernel A/B runs)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("ARCLE_BENCH_FORCE_DIST"):  # the env var exercises the N>1 code path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        torch.cuda.set_device(0)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device(f"cuda:{local_rank}")

    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import O2ARCv2Env
    from arcle_amd import actions

    n = a.envs_per_gpu
    # shard = contiguous global env ids [rank*n, (rank+1)*n); per-shard seeds keyed by rank
    batch = EnvBatch(n, H, W, -1, "o2arc", dev)
    batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    FL = batch.elide_flag  # ARCLE_STEP_ELIDE_SELECTED: what ARCVecEnv passes (state evolves only through the kernels)
    batch.set_tasks_padded(*make_tasks(n, 1000 + rank))
    batch.reset()
    K, Wm = a.steps, a.warmup
    bbox_np, op_np = make_actions(K + Wm, n, 2000 + rank)
    bbox = torch.from_numpy(bbox_np).to(dev)
    op = torch.from_numpy(op_np).to(dev)
    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream
    bptr = [bbox[i].data_ptr() for i in range(K + Wm)]
    optr = [op[i].data_ptr() for i in range(K + Wm)]

    def barrier():
        if dist is not None:
            dist.barrier()

    for i in range(Wm):  # untimed warm-up steps
        batch.step_bbox_ptr(bptr[i], optr[i], FL, sh)
    torch.cuda.synchronize(dev)
    # snapshot of the state the timed region starts from (replayed below for the byte accounting)
    snap = {k: v.clone() for k, v in batch.planes.items()}
    snap_rec, snap_cnt = batch.rec.clone(), batch.cnt.clone()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # ---- timed region: exactly K steps, bracketed by barrier + synchronize --------------------------
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record(stream)  # HIP events on the stream the kernel is launched on
    for i in range(Wm, Wm + K):
        batch.step_bbox_ptr(bptr[i], optr[i], FL, sh)
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_avg_s = ev0.elapsed_time(ev1) * 1e-3 / K  # K back-to-back launches of arcle_step_kernel
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    status = batch.status()
    assert status == 0, f"device status {status}"
    total_steps = K * n * world

    # ---- algorithmic bytes of exactly those K launches: restore the snapshot and replay them (untimed) with
    #      the kernel's per-env byte accounting switched on ------------------------------------------------
    roofline = None
    if rank == 0:
        for k, v in snap.items():
            batch.planes[k].copy_(v)
        batch.rec.copy_(snap_rec)
        batch.cnt.copy_(snap_cnt)
        batch.enable_accounting(True)
        batch.accounting(clear=True)
        for i in range(Wm, Wm + K):
            batch.step_bbox_ptr(bptr[i], optr[i], FL, sh)
        torch.cuda.synchronize(dev)
        nbytes, nsteps = batch.accounting(clear=True)
        batch.enable_accounting(False)
        per_launch_bytes = nbytes / K
        achieved = per_launch_bytes / kernel_avg_s
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):  # HBM bytes/launch from the rocprofv3 PMC passes of this same command
            pmc = json.load(open(pmc_path))
            traffic, traffic_src = pmc.get("hbm_bytes_per_launch"), pmc.get("source")
        roofline = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "arcle_step_kernel", "avg_launch_us": kernel_avg_s * 1e6,
                    "algorithmic_bytes_per_launch": per_launch_bytes,
                    "algorithmic_bytes_per_env_step": nbytes / max(nsteps, 1),
                    "frac_of_measured_copy_peak_6.29TBps": achieved / 6.29e12}

    if rank == 0:
        out = {
            "metric": "env-steps/sec (whole node), O2ARCv2Env 30x30, 8192 envs/GPU",
            "value": total_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: O2ARCv2Env 30x30, 8192 envs/GPU, full 35-op O2ARC table "
                                   "uniform, BBoxWrapper 5-tuples uniform (fused bbox ingress), max_trial=-1",
                       "envs_per_gpu": n, "global_envs": n * world, "grid": [H, W], "ingress": "bbox",
                       "parallelism": f"env-shard x{world} (no data-path collective)"},
            "roofline": roofline,
        }
        if world == 1 and not a.no_extras:
            out["extras"] = {"rollout": rollout_leg(batch, bbox, op, Wm, dev)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(1000)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
