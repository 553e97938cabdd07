#!/usr/bin/env python3
"""bench.py — env-steps/sec of the ARCLE hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config c3|c2|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
(`python bench.py --gpus N` without a launcher spawns the N ranks itself.)

A "step" is ONE pass of the hot path over one batch: a single launch of the step kernel that applies one (selection,
operation) action to every env of this GPU.  Tasks, state and the whole action stream are resident in HBM before the
timed region starts.  HEADLINE (`value`, `ms_per_step`, `roofline`) = the step()-per-call form: a region's K launches are K
`arcle_step_bbox` calls, none of which sees the next step's actions (the loop a(t+1) = policy(obs(t)) of the reference's
examples/example_bbox.py:13-15).  Since round 5 every such launch orders ITSELF (object operations to the waves that start first,
derived inside groups of 32 envs from the operations the launch is about to execute; scheduling only).  c3 times beside the headline —
same run, same event clock, top-level `forms` block — the same K launches enqueued by ONE `arcle_step_many` call (what
`ARCVecEnv.capture` records) and the K single-step calls with the dispatch order switched off (the plain instantiation: every wave steps
the env of its own slot) — the in-run A/B of the ordering (`--no-forms` skips that
leg).  Envs are independent, so N GPUs = N shards, no data-path collective (weak scaling).
OUTPUT: rank 0 prints ONE compact JSON line on stdout (< 3 KB, the LAST line; the contract fields + `forms`, `roofline`,
`cpu_baseline`, `sustained`, `legs_us_per_step`, `collective` for N > 1); the full record (every leg with its own roofline block,
per-region times) goes to stderr as one `BENCH_FULL {...}` line and to gpurun_out/bench_full_<config>_n<N>_k<K>.json.
Workloads (`config.workload` names the one that ran):
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
eagerly launched regions are clocked by the host.  (The K launches are captured once: every region replays the same K action
batches on an evolving state.)  Besides the contract fields the line carries
  roofline      of the step kernel, all bytes counted BY THE KERNEL in this run (accounting instantiation, region 0 replayed once,
                untimed): `achieved` / `frac` = algorithmic HBM bytes (planes semantically read + written by the
                executed op / mode + 56 B) / the average launch duration (HIP-event pair on the launch stream around the K
                back-to-back launches) / 8 TB/s; `traffic` / `frac_by_traffic` = the bytes of every global-memory access the kernel
                actually issued (elided writes excluded, row padding and re-reads included); `note_cache` = state bytes vs the 256 MiB
                Infinity Cache; `pmc_crosscheck` = the FETCH_SIZE / WRITE_SIZE figure recorded under profiles/ (calibrated there);
  extras        (N = 1, c3) the paths users call, each with `us_per_step_batch` and its own kernel-counted roofline block: vec_api
                (ARCVecEnv: Python loop / step_many / capture + replay), research_env (the paper's training step, rows rewritten in
                full / incrementally), mask_ingress (int8 / bit-packed masks), host_actions (records of a host-resident policy:
                zero-copy / one copy node / two), single_env (Gym class: step and transition latency), transition_rows (stateless
                batched transition), rollout, batch_sweep (32 768 / 65 536 / 131 072 envs: the out-of-cache fraction, library tables vs arcle_autotune), other_configs
                (c2 / c4 on one rank / c5 with their real bounds), big_grid (max_grid_size 64x64 / 127x127: the workgroup-per-env
                kernels, bytes modelled per op kind);
  cpu_baseline  (N=1) on this box's host, bounded samples of the same workload: the oracle's C restatement (1 thread /
                ALL host cores) and `numpy_step` = a plain-NumPy one-env-at-a-time step() loop with the reference's call
                structure (oracle/numpy_env.py; leaner than the reference itself, labelled so), 1 process / all cores;
  sustained     (N=1) the headline graph replayed back to back on a side thread for the ~15 s of the single-threaded CPU legs;
  collective    (N>1) backend, world size and the number of ranks an all-reduce of ones actually counted;
  multi         (N>1, c3) bounded legs of BASELINE configs[3] / [4] on the same process group, after the headline: c4 = 8192 envs/GPU,
                step + ONE packed all-gather per step (overlapped on a side stream / serial / the collective alone, GB/s), c5 = ARCEnv
                4096 envs/GPU FloodFill-heavy, sharded without a collective (multi_legs below; --no-multi skips them).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29e12 measured copy)
STEP_AUTORESET = 1
STEP_PACK_OBS = 256
STEP_ROWS_INCREMENTAL = 512


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
    n = 2048 if threads == 1 else max(8192, 256 * threads)  # (>= 256 envs per thread and parallel region)
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


def _host_capacity():
    """(schedulable CPUs, cgroup CPU quota in cores or None): what the container may use, which can be far less than what it sees."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return avail, quota


def _numpy_baseline(avail, before_fork=None, quota=None):
    """The plain-NumPy per-env step() loop (oracle/numpy_env.py): one process, then one process per host core — every worker runs
    against the same 5 s wall-clock budget, so the leg is bounded whatever the box really grants the container."""
    import multiprocessing as mp
    from oracle import numpy_env as NE
    done, sec = NE.run_chunk((11, 64, 150, 30, 30))  # ~10 k steps: a few tenths of a second per 10 k
    steps1 = max(50, int(150 * 6.0 / max(sec, 1e-3)))  # aim at ~6 s of single-process stepping
    done, sec = NE.run_chunk((12, 64, min(steps1, 4000), 30, 30))
    one = done / sec
    # one process per core the container may really use: every schedulable CPU, capped by the cgroup quota (256 CPUs visible / 16 granted
    # on the driver's boxes: 256 workers time-slicing 16 cores measure the scheduler)
    procs = max(1, min(avail, int(math.ceil(quota))) if quota else avail)
    per = max(20, int(one * 5.0 / 64))  # ~5 s per worker if it had a core to itself
    if before_fork is not None:
        before_fork()
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(NE.run_chunk, [(100 + i, 64, per, 30, 30, 5.0) for i in range(procs)], chunksize=1)
    wall = time.perf_counter() - t0
    return {"value": one, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "note": "a leaner NumPy port than the reference itself (which measured 36 k env-steps/s/core, BASELINE.md 2): NOT the reference's CPU path",
            "sample": f"64 envs x {done // 64} C3 steps, one Python step() per env, oracle/numpy_env.py",
            "all_cores": {"value": sum(d for d, _ in res) / max(max(s for _, s in res), 1e-9), "unit": "env-steps/s",
                          "cores": procs, "sample": f"{procs} processes x 64 envs, 5 s budget each (wall incl. fork {wall:.1f} s)"}}


class Sustained(threading.Thread):
    """The headline hipGraph replayed back to back on a side thread while rank 0 times the single-threaded CPU baselines (the
    oracle's C call releases the GIL): the SUSTAINED rate over ~15 s of wall clock next to the burst figure of the timed regions
    — and a GPU that an outside sampler (rocm-smi) can see busy.  Never part of `value`."""

    def __init__(self, graph, dev, steps_per_replay, us_per_replay):
        super().__init__(daemon=True)
        self.graph, self.dev, self.per = graph, dev, steps_per_replay
        self.reps = max(1, min(400, int(20000.0 / max(us_per_replay, 1.0))))  # ~20 ms of queued work between synchronisations
        self.halt = threading.Event()
        self.done, self.sec, self.err = 0, 0.0, None

    def run(self):
        try:
            torch.cuda.set_device(self.dev)
            st = torch.cuda.Stream(self.dev)
            with torch.cuda.stream(st):
                t0 = time.perf_counter()
                while not self.halt.is_set():
                    for _ in range(self.reps):
                        self.graph.replay()
                    st.synchronize()
                    self.done += self.reps * self.per
                self.sec = time.perf_counter() - t0
        except Exception as exc:  # noqa: BLE001 - reported, never fatal
            self.err = f"{type(exc).__name__}: {exc}"

    def result(self):
        if self.err or self.sec <= 0:
            return {"error": self.err or "did not run"}
        return {"value": self.done / self.sec, "unit": "env-steps/s", "seconds": round(self.sec, 2), "env_steps": self.done,
                "note": "host clock over back-to-back replays of the headline graph (incl. one synchronisation per ~20 ms), run on a side "
                        "thread during the single-threaded CPU-baseline legs"}


def cpu_baseline(seed, sustain=None):
    avail, quota = _host_capacity()
    if sustain is not None:
        sustain.start()
    v1, n1, s1 = _cpu_run(1, seed, 10.0)

    def stop_sustain():  # (stopped before the fork pool and the all-cores leg: they want every core)
        if sustain is not None:
            sustain.halt.set()
            sustain.join(30.0)
    numpy_step = _numpy_baseline(avail, stop_sustain, quota)
    # all host cores: OpenMP over envs with every schedulable CPU — and, because a container may see far more CPUs than it is granted
    # (256 visible here; with 256 threads the same code ran SLOWER than with one), a short scan of smaller teams; `all_cores` is the
    # best of them with the thread count it really used, the scan is kept beside it
    scan, t = [], max(1, avail)
    while t >= 4 and len(scan) < 6:
        v, nt, st = _cpu_run(t, seed, 6.0 if t == avail else 2.0)
        scan.append({"threads": t, "value": v, "sample": f"{nt} envs x {st} steps"})
        t //= 2
    best = max(scan, key=lambda e: e["value"]) if scan else {"threads": 1, "value": v1, "sample": ""}
    return {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n1} envs x {s1} steps of the same C3 action stream, oracle/arcle_oracle.c, 1 thread",
            "host_cores_available": avail, "cgroup_cpu_quota_cores": quota,
            "all_cores": {"value": best["value"], "unit": "env-steps/s", "cores": best["threads"],
                          "sample": best["sample"] + ", same code, OpenMP over envs; best of the thread-count scan",
                          "with_every_visible_cpu": scan[0] if scan else None, "scan": scan},
            "numpy_step": numpy_step,
            "reference_note": "the reference itself cannot travel to this box; survey container: 36 k env-steps/s/core "
                              "(Xeon 2.10 GHz, BASELINE.md §2)"}


def _r(x, nd=4):
    """Rounds floats for the compact line (4 significant digits)."""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}")
    return x


def emit(out, a):
    """stdout gets ONE compact JSON line (< 3 KB: the driver keeps an 8 KB tail and parses the last line); the full record — every
    leg with its own roofline block, the per-region times — goes to stderr as one `BENCH_FULL ` line and, when the directory is
    writable, to gpurun_out/bench_full.json."""
    full = json.dumps(out)
    print("BENCH_FULL " + full, file=sys.stderr, flush=True)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_full_{a.config}_n{out['n_gpus']}_k{out['steps']}.json"), "w") as f:
            f.write(full + "\n")
    except OSError:
        pass
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data")}
    c["value"], c["ms_per_step"] = _r(out["value"], 6), _r(out["ms_per_step"], 6)
    cfg = out["config"]
    c["config"] = {"workload": cfg["workload"][:150], "id": cfg["id"], "envs_per_gpu": cfg["envs_per_gpu"], "global_envs": cfg["global_envs"],
                   "grid": cfg["grid"], "ingress": cfg["ingress"], "parallelism": cfg["parallelism"][:90]}
    c["headline_form"] = "per_step_calls"
    t = out["timing"]
    c["timing"] = {"regions": t["regions"], "stat": "median region, max over ranks", "clock": "hip_events" if t["clock"].startswith("HIP") else "host",
                   "launch": "hipGraph of K arcle_step_bbox calls" if t["launch"] != "eager" else "eager",
                   "host_region_ms_median": _r(float(np.median(t["host_region_ms"])))}
    if out.get("forms"):
        c["forms"] = {k: {"value": _r(v["value"], 6), "avg_launch_us": _r(v["avg_launch_us"]), "frac": _r(v.get("frac"))} for k, v in out["forms"].items()}
        c["forms"]["note"] = "same run, same clock: one arcle_step_many call / single-step calls with arcle_set_dispatch_order(env, 0)"
    rl = out.get("roofline")
    if rl:
        c["roofline"] = {k: _r(rl[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_by_traffic", "kernel", "avg_launch_us",
                                                 "algorithmic_bytes_per_launch", "algorithmic_bytes_per_env_step") if k in rl}
        c["roofline"]["kernel"] = str(rl["kernel"])[:70]
        if rl.get("plan"):
            c["roofline"]["plan"] = rl["plan"]
        c["roofline"]["traffic_source"] = "kernel-counted issued bytes, this run"
        c["roofline"]["fits_infinity_cache"] = rl["note_cache"]["fits_infinity_cache"]
        if rl.get("pmc_crosscheck"):
            c["roofline"]["pmc_hbm_bytes_per_launch"] = rl["pmc_crosscheck"]["hbm_bytes_per_launch"]
    ex = out.get("extras") or {}
    sweep = ex.get("batch_sweep")
    if isinstance(sweep, list) and rl:
        def plan_str(pl):
            return ("grouped" if pl["orders_itself"] else (pl["policy"] or "plain")) + f"/{pl['waves_per_workgroup']}w" + ("*" if pl["autotuned"] else "")
        c["roofline"]["frac_out_of_cache"] = {str(e["envs"]): {"frac": _r(e["roofline"]["frac"]), "frac_by_traffic": _r(e["roofline"]["frac_by_traffic"]),
                                                              "us": _r(e["us_per_step_batch"]), "plan": plan_str(e["roofline"]["plan"]),
                                                              "table_plan_us": _r(e["table_plan"]["us_per_step_batch"])} for e in sweep}
    if ex:
        def us(*path):
            d = ex
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return _r(d) if isinstance(d, float) else ("err" if isinstance(ex.get(path[0]), dict) and "error" in ex[path[0]] else None)
        c["legs_us_per_step"] = {"vec_api_capture": us("vec_api", "us_per_step_batch"), "vec_api_python_loop": us("vec_api", "python_loop_us"),
                                 "research_env": us("research_env", "us_per_step_batch"), "mask_int8": us("mask_ingress", "mask", "us_per_step_batch"),
                                 "mask_bits": us("mask_ingress", "bits", "us_per_step_batch"), "host_actions": us("host_actions", "us_per_step_batch"),
                                 "single_env_step": us("single_env", "us_per_step"), "transition_rows": us("transition_rows", "us_per_step_batch"),
                                 "rollout": us("rollout", "us_per_step_batch"), "c2": us("other_configs", "c2", "us_per_step_batch"),
                                 "c4": us("other_configs", "c4", "us_per_step_batch"), "c5": us("other_configs", "c5", "us_per_step_batch"),
                                 "big_64x64_4096": us("big_grid", "64x64_4096", "us_per_step_batch"),
                                 "big_40x40_16384": us("big_grid", "40x40_16384", "us_per_step_batch"),
                                 "big_40x40_16384_bits": us("big_grid", "40x40_16384_bits", "us_per_step_batch")}
    cb = out.get("cpu_baseline")
    if cb:
        ns = cb["numpy_step"]
        c["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:110],
                             "host_cores_available": cb["host_cores_available"], "cgroup_cpu_quota_cores": cb.get("cgroup_cpu_quota_cores"),
                             "all_cores": {"value": _r(cb["all_cores"]["value"]), "cores": cb["all_cores"]["cores"],
                                           "with_every_visible_cpu": _r((cb["all_cores"].get("with_every_visible_cpu") or {}).get("value"))},
                             "numpy_step": {"value": _r(ns["value"]), "cores": 1, "all_cores": {"value": _r(ns["all_cores"]["value"]), "cores": ns["all_cores"]["cores"]},
                                            "note": "leaner NumPy port, not the reference's own step (36 k/s/core, BASELINE.md)"}}
    if out.get("sustained"):
        c["sustained"] = {k: _r(v, 6) for k, v in out["sustained"].items() if k in ("value", "seconds", "error")}
    for k in ("collective", "floodfill"):
        if k in out:
            c[k] = out[k]
    m = out.get("multi")
    if m:  # (N > 1: BASELINE configs[3] / [4] on the same process group)
        c["multi"] = {"error": m["error"][:160]} if "error" in m else {
            "c4_us_per_step": _r(m["c4"]["us_per_step"]), "c4_serial_us_per_step": _r(m["c4"]["serial_us_per_step"]),
            "c4_gather_only_us": _r(m["c4"]["gather_only_us"]), "c4_value": _r(m["c4"]["value"], 6), "c4_global_envs": m["c4"]["global_envs"],
            "gather_GBps": _r(m["c4"]["gather_GBps"]), "gather_GBps_per_link_dir": _r(m["c4"]["gather_GBps_per_link_dir"]),
            "c5_us_per_step": _r(m["c5"]["us_per_step"]), "c5_value": _r(m["c5"]["value"], 6), "c5_global_envs": m["c5"]["global_envs"],
            "transport": m["c4"]["transport"][:40], "steps_per_region": m["steps_per_region"], "regions": m["regions"]}
    c["full_record"] = "stderr line BENCH_FULL / gpurun_out/bench_full_*.json"
    line = json.dumps(c)
    if len(line) > 3800:  # never let the headline grow past what the driver keeps
        for k in ("legs_us_per_step", "sustained", "floodfill"):
            c.pop(k, None)
        line = json.dumps(c)
    print(line, flush=True)


# ---------------------------------------------------------------------------------------------------------------
# helpers shared by the legs: graph-replayed timing, kernel-counted bytes, a roofline block
# ---------------------------------------------------------------------------------------------------------------
def graph_time(dev, enqueue, K, reps=5, warm=3):
    """Captures `enqueue(stream_handle)` (K step launches) into one hipGraph, replays it `reps` times, returns the median seconds
    per step (HIP events around each replay, recorded on the stream the replay is launched on)."""
    st = torch.cuda.Stream(dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        enqueue(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(warm):
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
    return float(np.median(ts)), g


def counted_bytes(batch, enqueue, K, dev):
    """The K launches of `enqueue` replayed from the batch's CURRENT state (restored afterwards) with the kernel's byte accounting on:
    -> (algorithmic bytes per launch — SURVEY.md 8d, the roofline numerator —, bytes of the accesses the kernel actually issued per
    launch, algorithmic bytes per env-step)."""
    snap = batch.get_state()
    torch.cuda.synchronize(dev)
    batch.enable_accounting(True)
    batch.accounting_ex(clear=True)
    enqueue(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    alg, issued, steps = batch.accounting_ex(clear=True)
    batch.enable_accounting(False)
    batch.set_state(snap)
    torch.cuda.synchronize(dev)
    return alg / K, issued / K, alg / max(steps, 1)


def roofline_block(kernel, sec, alg, issued, n, PS=1024, planes=8, note=None, plan=None):
    """The JSON block of one kernel: achieved = algorithmic bytes / launch duration against the 8 TB/s HBM3E peak; `traffic` = the
    bytes of the accesses the kernel itself counted as issued (kernel-counted, this run)."""
    state = planes * n * PS + 24 * n
    blk = {"bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": alg / sec / HBM_PEAK,
           "traffic": issued, "traffic_source": "kernel-counted in this run (arcle_get_accounting_ex: every global-memory access the "
           "step kernel issued, whole 16-byte lanes incl. row padding; elided writes excluded, re-reads included)",
           "frac_by_traffic": issued / sec / HBM_PEAK, "kernel": kernel, "avg_launch_us": sec * 1e6,
           "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_env_step": alg / n,
           "frac_of_measured_copy_peak_6.29TBps": alg / sec / 6.29e12,
           "note_cache": {"state_bytes": state, "infinity_cache_bytes": 256 << 20,
                          "fits_infinity_cache": state < (256 << 20),
                          "meaning": "when the state fits the 256 MiB Infinity Cache the bytes are moved, but not all of them reach "
                                     "HBM: frac is then a fabric-level figure; extras.batch_sweep holds the out-of-cache fractions"}}
    if note:
        blk["note"] = note
    if plan is not None:  # how the launches ran: arcle_launch_info (self-ordering, cache policy of the speculative grid request, workgroup size)
        blk["plan"] = plan
    return blk


def make_batch(dev, n, seed=1000, kind="o2arc", H=30, W=30):
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv, O2ARCv2Env
    batch = EnvBatch(n, H, W, -1, kind, dev)
    cls = ARCEnv if kind == "arc" else O2ARCv2Env
    batch.set_op_table(actions.table_descs(cls.default_operations()))
    batch.set_tasks_padded(*make_tasks(n, seed, H, W))
    batch.reset()
    return batch


# ---------------------------------------------------------------------------------------------------------------
# non-headline legs (extras): every one carries us_per_step_batch and — where a step kernel is what is timed — a roofline block
# ---------------------------------------------------------------------------------------------------------------
def rollout_leg(batch, bbox, op, dev, T=128, reps=8):
    """NOT the headline metric: the same action stream replayed with arcle_rollout_bbox (T steps per launch, env state
    resident in registers between the steps; only per-step reward/terminated and the final state reach HBM)."""
    T = min(T, bbox.shape[0])
    bb, oo = bbox[:T].contiguous(), op[:T].contiguous()
    FL = batch.elide_flag | STEP_AUTORESET  # (what ARCVecEnv.rollout_bbox passes: the lean instantiation with the flags as constants)

    def timed(**kw):
        batch.rollout(bb, oo, FL, **kw)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            batch.rollout(bb, oo, FL, **kw)
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / reps
    sec = timed()
    rows = torch.empty((T, batch.N, batch.packed_obs_size()), dtype=torch.uint8, device=dev)
    sec_rows = timed(packed=rows)  # ... and with the packed observation row of every step (ARCLE_STEP_PACK_OBS: 912 B per env-step out)
    return {"mode": "arcle_rollout_bbox", "steps_per_launch": T, "value": T * batch.N / sec, "unit": "env-steps/s",
            "us_per_step_batch": sec / T * 1e6, "with_packed_rows_us_per_step_batch": sec_rows / T * 1e6,
            "note": "state stays on chip between steps; not comparable with the per-step HBM roofline above"}


def vec_api_leg(dev, n, bbox, op, K=100):
    """The product's front-end, ARCVecEnv, driven the three ways it offers: one Python call per step (`step_bbox`), K steps per call
    into the library (`step_many`), K steps captured once and replayed (`capture` / `replay`: the action buffers are re-read at
    replay time).  `value` = the captured form — what a training loop uses."""
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    v = ARCVecEnv(O2ARCv2Env, n, SyntheticLoader(n_tasks=400, seed=1, max_size=(30, 30)), device=dev, seed=7, autoreset=True)
    v.reset()
    K = min(K, bbox.shape[0])
    bb, oo = bbox[:K].contiguous(), op[:K].contiguous()
    out = {}
    for i in range(10):
        v.step_bbox(bb[i], oo[i])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(K):
        v.step_bbox(bb[i], oo[i])
    torch.cuda.synchronize(dev)
    out["python_loop_us"] = (time.perf_counter() - t0) / K * 1e6
    v.step_many(bb, oo)
    torch.cuda.synchronize(dev)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        v.step_many(bb, oo)
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) / K * 1e6)
    out["step_many_us"] = float(np.median(ts))
    cs = v.capture(bb, oo)
    for _ in range(3):
        cs.replay()
    torch.cuda.synchronize(dev)
    ts, ds = [], []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        cs.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) / K * 1e6)
        ds.append(e0.elapsed_time(e1) * 1e-3 / K)
    sec_host = float(np.median(ts)) * 1e-6  # host clock around replay + synchronize: what the caller's loop sees
    b = v.batch
    alg, issued, _ = counted_bytes(b, lambda sh: [b.step_bbox_ptr(bb[i].data_ptr(), oo[i].data_ptr(), v.flags, sh) for i in range(K)], K, dev)
    v.check_errors()
    out.update({"mode": f"ARCVecEnv.capture({K} steps) + replay(), host clock incl. the synchronisation", "value": n / sec_host,
                "unit": "env-steps/s", "us_per_step_batch": sec_host * 1e6, "device_us_per_step_batch": float(np.median(ds)) * 1e6,
                "steps_per_call": K, "roofline": roofline_block("arcle_step_kernel<bbox, FULL, 0, 0, autoreset|elide, 30>", sec_host, alg, issued, n)})
    return out


def research_env_leg(dev, n, bbox, op, K=200):
    """The step the reference's training script runs (agents/env.py + agents/train.py:61-68) — op 33 = crop, dense reward,
    TimeLimit(100) truncation, next-step autoreset onto a NEW device-drawn task with colour-permutation + rot90 augmentation, and
    the FilterO2ARC + FlattenObservation row of every env written by the step kernel — all in ONE launch per step, graph-replayed."""
    from arcle_amd import actions
    from arcle_amd.engine import STEP_FLAT_OBS
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader

    class Crop(O2ARCv2Env):  # agents/env.py:23-28
        def create_operations(self):
            ops = super().create_operations()
            ops[33] = actions.reset_sel(actions.crop_grid)
            return ops
    limit = int(os.environ.get("ARCLE_BENCH_RESEARCH_LIMIT", "100"))  # (tuning runs: how much of the step its auto-resets are)
    v = ARCVecEnv(Crop, n, SyntheticLoader(n_tasks=400, seed=1, max_size=(30, 30)), device=dev, seed=7, autoreset="resample", augment=("permute", "rot90"),
                  dense_reward=True, max_episode_steps=limit)
    v.reset()
    rows = v.enable_flat_rows(filtered=True)  # a live [N, 2710] mirror of the FilterO2ARC rows, kept by the step kernel
    K = min(K, bbox.shape[0])
    b = v.batch
    # desynchronise the episodes first (all envs start at step 0: left alone, every env would hit TimeLimit in the same launch —
    # a training run is never in that state after its first episode)
    b.cnt[:, 0] = torch.randint(0, min(limit, 100), (n,), device=dev, dtype=torch.int32)
    for i in range(100):
        b.step_bbox_ptr(bbox[i % K].data_ptr(), op[(i * 7 + 3) % K].data_ptr(), v.flags, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    out = {}
    for name, FL in (("rows_rewritten_in_full", v.flags & ~STEP_ROWS_INCREMENTAL), ("rows_incremental", v.flags)):
        def enqueue(sh, FL=FL):
            for i in range(K):
                b.step_bbox_ptr(bbox[i].data_ptr(), op[i].data_ptr(), FL, sh)
        alg, issued, _ = counted_bytes(b, enqueue, K, dev)
        v._refresh_rows()  # (the byte count replayed the steps and restored the state: bring the mirrored rows back in line)
        sec, _ = graph_time(dev, enqueue, K)
        out[name] = {"value": n / sec, "unit": "env-steps/s", "us_per_step_batch": sec * 1e6,
                     "roofline": roofline_block("arcle_step_kernel<bbox, FULL, 0, 1, research flags, 30>", sec, alg, issued, n,
                                                note="algorithmic = the step's planes + 56 B, + the dense reward's answer read when the grid moved, "
                                                     "+ per row: its three planes read once and 2710 B written (SURVEY.md 8d style, also for the "
                                                     "incremental writer, which leaves unchanged segments alone: compare `traffic`); the kernel re-reads "
                                                     "the planes it just stored through its own L1/L2, so issued > HBM traffic for the full rewrite")}
    assert b.status() == 0
    assert torch.equal(rows, b.flat_obs(filtered=True)), "the incrementally kept rows drifted from the state"
    best = out["rows_incremental"]
    out.update({"mode": "ARCVecEnv(autoreset='resample', augment, dense_reward, max_episode_steps=100).enable_flat_rows(filtered=True): "
                        "one launch per step keeps the FilterO2ARC rows of all envs current (incremental writer); the full-rewrite form beside it",
                "value": best["value"], "unit": "env-steps/s", "us_per_step_batch": best["us_per_step_batch"], "row_bytes": int(rows.shape[1]),
                "roofline": best["roofline"]})
    return out


def ingress_leg(dev, n, bbox, op, K=64):
    """The reference's native action type is a full H x W mask (base.py:134-138).  Same rectangles as the headline's bbox tuples, sent
    as int8 masks (arcle_step_mask) and as bit-packed boolean masks (arcle_step_bits, 128 B per env)."""
    K = min(K, bbox.shape[0])
    bb = bbox[:K]
    x1, x2 = torch.minimum(bb[..., 0], bb[..., 2]), torch.maximum(bb[..., 0], bb[..., 2])
    y1, y2 = torch.minimum(bb[..., 1], bb[..., 3]), torch.maximum(bb[..., 1], bb[..., 3])
    ii = torch.arange(30, device=dev)[None, None, :, None]
    jj = torch.arange(30, device=dev)[None, None, None, :]
    masks = ((ii >= x1[..., None, None]) & (ii <= x2[..., None, None]) & (jj >= y1[..., None, None]) & (jj <= y2[..., None, None])).to(torch.int8).contiguous()
    out = {}
    for form in ("mask", "bits"):
        batch = make_batch(dev, n)
        FL = batch.elide_flag | STEP_AUTORESET
        if form == "bits":
            pay = torch.stack([batch.pack_mask_bits(masks[i]) for i in range(K)])
            fn = batch.L.arcle_step_bits
        else:
            pay, fn = masks, batch.L.arcle_step_mask

        def enqueue(sh, pay=pay, fn=fn, batch=batch, FL=FL):
            for i in range(K):
                rc = fn(batch._h, pay[i].data_ptr(), op[i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh)
                assert rc == 0
        alg, issued, _ = counted_bytes(batch, enqueue, K, dev)
        sec, _ = graph_time(dev, enqueue, K)

        def enqueue_warm(sh, pay=pay, fn=fn, batch=batch, FL=FL):  # the same launches over 8 payload batches: payload + state stay inside the
            for i in range(K):                                      # Infinity Cache — masks a policy wrote on the device just before the step
                rc = fn(batch._h, pay[i & 7].data_ptr(), op[i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh)
                assert rc == 0
        sec_warm, _ = graph_time(dev, enqueue_warm, K)
        out[form + "_payload_cache_resident_us"] = sec_warm * 1e6
        out[form] = {"value": n / sec, "unit": "env-steps/s", "us_per_step_batch": sec * 1e6,
                     "payload_bytes_per_env": 900 if form == "mask" else 128,
                     "roofline": roofline_block(f"arcle_step_kernel<{form}, FULL, 0, 0, autoreset|elide{'|grouped' if batch.orders_itself(form, FL) else ''}, 30>",
                                                sec, alg, issued, n, plan=batch.launch_info(form, FL))}
    batch = make_batch(dev, n)
    sec, _ = graph_time(dev, lambda sh: [batch.pack_mask_bits(masks[i], pay[i]) for i in range(K)], K)
    out["pack_mask_bits_us"] = sec * 1e6
    out["mode"] = ("full H x W selection masks: int8 [N,900] / bit-packed [N,128]; headline figures with 64 distinct payload batches (472 MB of int8 "
                   "masks: every launch streams its 7.4 MB from HBM), *_payload_cache_resident_us with 8")
    out["value"], out["unit"], out["us_per_step_batch"] = out["mask"]["value"], "env-steps/s", out["mask"]["us_per_step_batch"]
    return out


def host_actions_leg(dev, n, bbox, op, K=200):
    """PCIe-inclusive rates of a HOST-resident policy (never part of `value`): the action of every env is the BBoxWrapper 5-tuple
    record (20 B per env, arcle_step_bbox5).  (a) arcle_step_many over the host array: eight extra workgroups at the front of launch
    t copy step t+1's records into a device staging buffer while launch t runs; (b) zero-copy: every wave reads its own record
    straight from pinned host memory; (c) one copy node (pinned host -> device) in front of every step; (d) round 2's form: two
    arrays, two copy nodes."""
    K = min(K, bbox.shape[0])
    act5 = torch.cat([bbox[:K], op[:K, :, None]], -1).contiguous()
    h5 = act5.cpu().pin_memory()
    hb, ho = bbox[:K].cpu().pin_memory(), op[:K].cpu().pin_memory()
    out = {}
    batch = make_batch(dev, n)
    FL = batch.elide_flag | STEP_AUTORESET
    L, h = batch.L, batch._h

    def zero_copy(sh):
        for i in range(K):
            assert L.arcle_step_bbox5(h, h5[i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh) == 0
    d5 = torch.empty_like(act5[0])

    def one_copy(sh):
        for i in range(K):
            d5.copy_(h5[i], non_blocking=True)
            assert L.arcle_step_bbox5(h, d5.data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh) == 0
    db, do = torch.empty_like(bbox[0]), torch.empty_like(op[0])

    def two_copies(sh):
        for i in range(K):
            db.copy_(hb[i], non_blocking=True)
            do.copy_(ho[i], non_blocking=True)
            batch.step_bbox_ptr(db.data_ptr(), do.data_ptr(), FL, sh)
    rw, tm = torch.empty((K, n), dtype=torch.int32, device=dev), torch.empty((K, n), dtype=torch.uint8, device=dev)

    def prefetched(sh):  # ONE library call for the K steps over the host array: the front workgroups of launch t fetch step t+1's records
        assert L.arcle_step_many(h, 3, K, h5.data_ptr(), None, rw.data_ptr(), tm.data_ptr(), FL, sh) == 0
    prefetched(torch.cuda.current_stream(dev).cuda_stream)  # (eagerly once: the library allocates its staging buffer outside a capture)
    torch.cuda.synchronize(dev)
    for name, fn in (("prefetched_by_the_previous_launch", prefetched), ("zero_copy_pinned_records", zero_copy),
                     ("one_copy_node_records", one_copy), ("two_copy_nodes_round2", two_copies)):
        sec, _ = graph_time(dev, fn, K)
        out[name] = {"us_per_step_batch": sec * 1e6, "value": n / sec}
    alg, issued, _ = counted_bytes(batch, zero_copy, K, dev)
    best = min(("prefetched_by_the_previous_launch", "zero_copy_pinned_records", "one_copy_node_records"), key=lambda k: out[k]["us_per_step_batch"])
    sec = out[best]["us_per_step_batch"] * 1e-6
    out.update({"mode": f"BBoxWrapper 5-tuple records from pinned host memory ({best})", "value": n / sec, "unit": "env-steps/s",
                "us_per_step_batch": sec * 1e6, "host_bytes_per_step": int(n * 20),
                "roofline": roofline_block("arcle_step_kernel<bbox5, FULL, 0, 0, autoreset|elide, 30>", sec, alg, issued, n,
                                           note="20 B per env of the issued bytes cross PCIe, not HBM")})
    return out


def single_env_leg(dev, reps=300):
    """The single-env drop-in class (the reference's Gym API: numpy state dict on the host after every step): latency of step() and
    of transition(deepcopy(state), action).  One env on a GPU is latency-bound by construction; the reference's CPU step: 27.6 us."""
    import copy
    from arcle_amd.envs import O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    env = O2ARCv2Env(data_loader=SyntheticLoader(n_tasks=20, seed=1), max_grid_size=(30, 30), device=dev)
    obs, info = env.reset()
    rng = np.random.default_rng(0)
    acts = []
    for _ in range(reps):
        sel = np.zeros((30, 30), np.int8)
        x, y = rng.integers(0, 25, 2)
        sel[x:x + rng.integers(1, 5), y:y + rng.integers(1, 5)] = 1
        acts.append({"selection": sel, "operation": int(rng.integers(0, 34))})
    for a in acts[:20]:
        env.step(a)
    t0 = time.perf_counter()
    for a in acts:
        obs, r, term, trunc, info = env.step(a)
    step_us = (time.perf_counter() - t0) / len(acts) * 1e6
    st = copy.deepcopy(obs)
    for a in acts[:10]:
        env.transition(st, a)
    t0 = time.perf_counter()
    for a in acts[:100]:
        env.transition(st, a)
    tr_us = (time.perf_counter() - t0) / 100 * 1e6
    return {"mode": "O2ARCv2Env.step / .transition, one env, numpy state dict on the host (ONE launch + ONE synchronisation per call; "
                    "action and state row live in pinned host memory the kernel reads / writes directly)",
            "us_per_step": step_us, "us_per_transition": tr_us, "value": 1e6 / step_us, "unit": "env-steps/s",
            "us_per_step_batch": step_us, "reference_cpu_us_per_step": 27.6}


def transition_leg(dev, n, bbox, op, K=32):
    """The stateless batched transition (arcle_transition_rows): n (state row, action) pairs per launch, rows in and out in HBM,
    no resident env touched — the planning / search primitive of README.md:55."""
    batch = make_batch(dev, n)
    FL = batch.elide_flag | STEP_AUTORESET
    for i in range(20):
        batch.step_bbox_ptr(bbox[i].data_ptr(), op[i].data_ptr(), FL, torch.cuda.current_stream(dev).cuda_stream)
    rows = batch.get_state_rows()  # (as the library hands rows out: a [n, L] view of a 16-byte aligned buffer, stride L rounded up to 16)
    stride = rows.stride(0)
    out = torch.empty((n, ((batch.state_row_size() + 15) & ~15)), dtype=torch.int8, device=dev)
    rw, tm = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    K = min(K, bbox.shape[0])

    def enqueue(sh):
        for i in range(K):
            rc = batch.L.arcle_transition_rows(batch._h, n, rows.data_ptr(), stride, 1, bbox[i].data_ptr(), op[i].data_ptr(), None,
                                               out.data_ptr(), out.shape[1], 0, rw.data_ptr(), tm.data_ptr(), 0, sh)
            assert rc == 0
    sec, _ = graph_time(dev, enqueue, K)
    L = batch.state_row_size()
    moved = n * (2 * L + 1024 + 16 + 24)
    # ... and from densely packed rows (stride L, no alignment: a contiguous clone) — those cannot be staged chunk-wise and take the on-demand form
    rows_d = rows.clone()

    def enqueue_dense(sh):
        for i in range(K):
            rc = batch.L.arcle_transition_rows(batch._h, n, rows_d.data_ptr(), rows_d.stride(0), 1, bbox[i].data_ptr(), op[i].data_ptr(), None,
                                               out.data_ptr(), out.shape[1], 0, rw.data_ptr(), tm.data_ptr(), 0, sh)
            assert rc == 0
    sec_d, _ = graph_time(dev, enqueue_dense, K)
    # in place: rows_out IS rows_in — the kernel rewrites only the planes the op changed (walking n trajectories forward)
    buf = torch.zeros((n, out.shape[1]), dtype=torch.int8, device=dev)
    buf[:, :L] = rows

    def enqueue_in_place(sh):
        for i in range(K):
            rc = batch.L.arcle_transition_rows(batch._h, n, buf.data_ptr(), buf.shape[1], 1, bbox[i].data_ptr(), op[i].data_ptr(), None,
                                               buf.data_ptr(), buf.shape[1], 0, rw.data_ptr(), tm.data_ptr(), 0, sh)
            assert rc == 0
    sec_ip, _ = graph_time(dev, enqueue_in_place, K)
    return {"mode": f"arcle_transition_rows, {n} (row, bbox action) pairs per launch, rows {L} B", "value": n / sec, "unit": "transitions/s",
            "us_per_step_batch": sec * 1e6,
            "rows_densely_packed": {"us_per_step_batch": sec_d * 1e6, "note": "input rows with stride L (unaligned): the on-demand form, as in round 4"},
            "in_place": {"us_per_step_batch": sec_ip * 1e6, "value": n / sec_ip, "unit": "transitions/s",
                         "note": "rows_out is rows_in: only the planes the op changed are rewritten"},
            "roofline": {"bound": "hbm", "achieved": moved / sec / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": moved / sec / HBM_PEAK,
                         "traffic": None, "algorithmic_bytes_per_launch": moved,
                         "note": "algorithmic = row in + row out + answer plane + action/outputs per pair (every out-of-place transition reads and "
                                 "writes the WHOLE 6314-byte state: it has no resident copy to leave untouched planes in)"}}


def batch_sweep_leg(dev, bbox, op, sizes=(32768, 65536, 131072), K=24):
    """The headline kernel on batches whose state (8 planes x N x 1024 B) exceeds the 256 MiB Infinity Cache: the out-of-cache
    HBM fraction, timed in this run — with the launch plan of the library's tables, and after `arcle_autotune` picked the plan on this
    handle, this box and these action arrays (the in-box A/B of the policy choice: `autotune.candidates` lists every plan it timed)."""
    out = []
    for N in sizes:
        rep = (N + bbox.shape[1] - 1) // bbox.shape[1]
        bb = bbox[:K].repeat(1, rep, 1)[:, :N].contiguous()
        oo = op[:K].repeat(1, rep)[:, :N].contiguous()
        batch = make_batch(dev, N, seed=1234)
        FL = batch.elide_flag | STEP_AUTORESET

        def enqueue(sh, batch=batch, bb=bb, oo=oo, FL=FL):
            for i in range(K):
                batch.step_bbox_ptr(bb[i].data_ptr(), oo[i].data_ptr(), FL, sh)
        alg, issued, _ = counted_bytes(batch, enqueue, K, dev)
        plan0 = batch.launch_info("bbox", FL)
        sec0, _ = graph_time(dev, enqueue, K, reps=7, warm=8)  # (1 GB of freshly allocated state: let TLBs and clocks settle)
        cands = batch.autotune("bbox", bb, oo, FL)  # (on the leg's own K action batches)
        plan1 = batch.launch_info("bbox", FL)
        sec1, _ = graph_time(dev, enqueue, K, reps=7, warm=4)
        sec, plan = (sec1, plan1) if sec1 <= sec0 else (sec0, plan0)
        leg = {"envs": N, "us_per_step_batch": sec * 1e6, "value": N / sec, "unit": "env-steps/s",
               "roofline": roofline_block("arcle_step_kernel<bbox, FULL, 0, 0, autoreset|elide, 30>", sec, alg, issued, N, plan=plan),
               "table_plan": {"plan": plan0, "us_per_step_batch": sec0 * 1e6}, "autotuned_plan": {"plan": plan1, "us_per_step_batch": sec1 * 1e6},
               "autotune": {"candidates": [f"{'grouped' if r['orders_itself'] else (r['policy'] or 'plain')}/{r['waves_per_workgroup']}w {r['us_per_launch']:.2f}us" for r in cands],
                            "note": "arcle_autotune: every candidate walks the leg's K action batches from the saved state; the leg's own graph is "
                                    "then timed with the table's plan and with the tuned plan, the faster one is the leg's figure"}}
        out.append(leg)
        del batch, bb, oo
        torch.cuda.empty_cache()
    return out


def other_configs_leg(dev, K=100):
    """The other single-GPU workloads of BASELINE.json, timed by the same command (graph-replayed, HIP events): c2 (10x10, 1024
    envs: one eighth of an occupancy round — bound by the launch floor, not by HBM), c4 on one rank (step + fused packed gather
    row), c5 (ARCEnv flood fills: bound by the fill's dependent passes)."""
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv, O2ARCv2Env
    out = {}
    for name in ("c2", "c4", "c5"):
        cfg = CONFIGS[name]
        H, W, n, kind = cfg["H"], cfg["W"], cfg["envs"], cfg["kind"]
        batch = EnvBatch(n, H, W, cfg["max_trial"], kind, dev)
        batch.set_op_table(actions.table_descs((ARCEnv if kind == "arc" else O2ARCv2Env).default_operations()))
        if name == "c5":
            tasks, (bb, oo) = make_tasks_c5(n, 1000, H, W), make_actions_c5(K, n, 2000, H, W)
        elif name == "c2":
            tasks, (bb, oo) = make_tasks(n, 1000, H, W, lo=3, zero_frac=0.5), make_actions_c2(K, n, 2000)
        else:
            tasks, (bb, oo) = make_tasks(n, 1000, H, W), make_actions(K, n, 2000, H, W)
        batch.set_tasks_padded(*tasks)
        batch.reset()
        FL = batch.elide_flag | cfg["flags"]
        if name == "c4":
            batch.set_packed_output()
            FL |= STEP_PACK_OBS
        bbd, ood = torch.from_numpy(bb).to(dev), torch.from_numpy(oo).to(dev)

        def enqueue(sh, batch=batch, bbd=bbd, ood=ood, FL=FL):
            for i in range(K):
                batch.step_bbox_ptr(bbd[i].data_ptr(), ood[i].data_ptr(), FL, sh)
        alg, issued, _ = counted_bytes(batch, enqueue, K, dev)
        sec, _ = graph_time(dev, enqueue, K)
        leg = {"workload": cfg["name"], "envs": n, "us_per_step_batch": sec * 1e6, "value": n / sec, "unit": "env-steps/s"}
        if name == "c4":  # the same single-step calls with the dispatch order off (the plain packed-row instantiation): the in-run A/B
            batch.set_dispatch_order(False)
            sec_p, _ = graph_time(dev, enqueue, K)
            batch.set_dispatch_order(True)
            leg["dispatch_order_off_us_per_step_batch"] = sec_p * 1e6
        rl = roofline_block("arcle_step_kernel", sec, alg, issued, n, PS=batch.PS, planes=len(batch.planes))
        if name == "c2":
            rl.update({"bound": "launch-floor", "launch_floor_us": 2.5,
                       "note": "1024 envs = 1024 waves = 1/8 of one occupancy round of the chip: the launch cannot be shorter than the "
                               "~2.5 us an EMPTY kernel of this shape takes (profiles/archive/round2_membench.txt); the HBM fraction is reported "
                               "for completeness and is not the bound"})
        if name == "c5":
            seeds = [(int(bb[0, e, 0]), int(bb[0, e, 1])) for e in range(n) if 10 <= oo[0, e] < 20]
            grids = [tasks[0][e] for e in range(n) if 10 <= oo[0, e] < 20]
            rl.update({"bound": "iteration (dependent flood-fill passes)", "floodfill_share_of_actions": float(((oo >= 10) & (oo < 20)).mean()),
                       "frontier_rounds": frontier_rounds_sample(grids, seeds),
                       "note": "a fill is a chain of dependent passes over the row board (a pass per corridor leg) on half an occupancy "
                               "round of waves; the HBM fraction is reported for completeness and is not the bound"})
        leg["roofline"] = rl
        out[name] = leg
        del batch
    return out


# planes a step of each op kind moves (reads + writes of whole PS-byte planes) on the workgroup-per-env path, the C3 table: the per-op
# model behind the big-grid leg's byte figure — an op that turns out to be a no-op (empty selection, inactive object) moves less
_BIG_PLANES = {"color": 3, "flood": 3, "move": 6, "rotflip": 8, "copy": 3, "paste": 4, "copy_from_input": 3, "reset_grid": 2, "resize_grid": 2, "submit": 2}


def _big_model_bytes(op, PS):
    k = np.select([op < 10, op < 20, op < 24, op < 28, op < 30, op == 30, op == 31, op == 32, op == 33],
                  [_BIG_PLANES["color"], _BIG_PLANES["flood"], _BIG_PLANES["move"], _BIG_PLANES["rotflip"], _BIG_PLANES["copy"],
                   _BIG_PLANES["paste"], _BIG_PLANES["copy_from_input"], _BIG_PLANES["reset_grid"], _BIG_PLANES["resize_grid"]], _BIG_PLANES["submit"])
    return float(k.sum()) * PS / op.shape[0] + 56.0 * op.shape[1]  # + record in / out, counters, action, outputs per env


def big_grid_case(dev, H, W, n, K=24, ops=None, ingress="bbox", point_seeds=False):
    """One max_grid_size beyond 1024 cells: K graph-replayed step launches of the C3 action mix (one workgroup per env).  `ingress`: "bbox"
    tuples (the headline's form), or the same rectangles as full int8 masks ("mask") / bit-packed boolean masks ("bits")."""
    batch = make_batch(dev, n, 1000, "o2arc", H, W)
    bb, oo = make_actions(K, n, 2000, H, W)
    if ops is not None:  # (tools/bigbench.py: one class of operations only)
        oo = (ops[0] + oo % (ops[1] - ops[0] + 1)).astype(np.int32)
    if point_seeds:  # (tools/bigbench.py --point-seeds: every selection one cell — FloodFill really fills, as with c5's seeds)
        bb[..., 2:] = bb[..., :2]
    bbd, ood = torch.from_numpy(bb).to(dev), torch.from_numpy(oo).to(dev)
    FL = batch.elide_flag | STEP_AUTORESET
    if ingress == "bbox":
        def enqueue(sh):
            for i in range(K):
                batch.step_bbox_ptr(bbd[i].data_ptr(), ood[i].data_ptr(), FL, sh)
    else:
        x1, x2 = torch.minimum(bbd[..., 0], bbd[..., 2]), torch.maximum(bbd[..., 0], bbd[..., 2])
        y1, y2 = torch.minimum(bbd[..., 1], bbd[..., 3]), torch.maximum(bbd[..., 1], bbd[..., 3])
        ii, jj = torch.arange(H, device=dev)[None, :, None], torch.arange(W, device=dev)[None, None, :]
        pay = []
        for i in range(K):  # (one step's masks at a time: 16 384 x 127 x 127 int8 = 264 MB)
            m = ((ii >= x1[i, :, None, None]) & (ii <= x2[i, :, None, None]) & (jj >= y1[i, :, None, None]) & (jj <= y2[i, :, None, None])).to(torch.int8).contiguous()
            pay.append(batch.pack_mask_bits(m) if ingress == "bits" else m)
        fn = batch.L.arcle_step_bits if ingress == "bits" else batch.L.arcle_step_mask

        def enqueue(sh):
            for i in range(K):
                rc = fn(batch._h, pay[i].data_ptr(), ood[i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh)
                assert rc == 0
    alg, issued, _ = counted_bytes(batch, enqueue, K, dev)
    sec, _ = graph_time(dev, enqueue, K)
    rl = roofline_block("arcle_big_step_lean<2, tuples>" if ingress == "bbox" else "arcle_big_step_lean<2, masks>", sec, alg, issued, n, PS=batch.PS, planes=len(batch.planes),
                        note="bytes counted by the generic kernel replaying the same launches in this run (every 16-byte access a thread issues; "
                             "`algorithmic` = the same without the row padding); bound: instructions issued per env (DESIGN.md §3), not HBM")
    rl["modelled_bytes_per_launch"] = _big_model_bytes(oo, batch.PS)  # (the per-op-kind model the first measurements of this path used)
    return {"workload": f"O2ARCv2Env {H}x{W}, {n} envs, C3 action mix, {ingress} ingress (max_grid_size beyond one wavefront: one workgroup per env)", "envs": n,
            "plane_stride": batch.PS, "us_per_step_batch": sec * 1e6, "value": n / sec, "unit": "env-steps/s", "roofline": rl}


def big_grid_leg(dev):
    """Grids of more than 1024 cells (the reference takes any max_grid_size, base.py:37-49): the workgroup-per-env kernels, timed like the
    other legs.  Not the headline and not ARC's regime (30 x 30) — the completeness path."""
    return {"64x64_4096": big_grid_case(dev, 64, 64, 4096), "127x127_1024": big_grid_case(dev, 127, 127, 1024),
            "40x40_16384": big_grid_case(dev, 40, 40, 16384), "40x40_16384_bits": big_grid_case(dev, 40, 40, 16384, K=12, ingress="bits")}


# ---------------------------------------------------------------------------------------------------------------
def multi_legs(dev, dist, world, rank, shared_gpu, K=20, R=3):
    """N > 1 only — after the c3 headline, on the same process group: bounded legs of BASELINE configs[3] and configs[4], so that the
    driver's plain `bench.py --gpus N` also measures what north_star calls the "RCCL-over-xGMI gather of (obs, reward, done)":

      c4  8192 envs/GPU (65 536 on 8), step with the fused packed-row epilogue + ONE all_gather_into_tensor of the 912-byte rows per step;
          `overlapped`: the collective on a side stream behind an event, rows double-buffered, step t+1 runs under the gather of step t
          (what ShardedVecEnv.gather_async does); `serial`: step, then the collective, on one stream; `gather_only`: the collective alone
      c5  ARCEnv, 4096 envs/GPU (32 768 on 8), 70 % FloodFill point seeds: sharded, no collective (one hipGraph of the K launches)

    Every figure is the max over ranks of the median of R regions of K steps (eager launches + host clock for c4 — the collective is
    30-40x longer than a launch; HIP events around a graph replay for c5).  Ranks sharing one GPU (functional runs on a one-GPU box)
    gather over gloo through a host bounce.  The sequence of collectives is the same on every rank whatever happens locally."""
    import torch.distributed  # noqa: F401
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64)
        t = t if shared_gpu else t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    res = {"steps_per_region": K, "regions": R, "clock": "host perf_counter between barrier + synchronize pairs (c4), HIP events around one graph replay (c5); median region, max over ranks"}
    # ---- setup on every rank, then ONE agreement collective before any data-path collective is issued --------------------------------
    err = None
    try:
        c4, c5 = CONFIGS["c4"], CONFIGS["c5"]
        n = c4["envs"]
        S = K + 4
        b4 = make_batch(dev, n, seed=1000 + rank)
        FL4 = b4.elide_flag | c4["flags"] | STEP_PACK_OBS
        bb_np, op_np = make_actions(S, n, 2000 + rank)
        bb4, op4 = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
        R_ = b4.packed_obs_size()
        packed = [torch.zeros((n, R_), dtype=torch.uint8, device=dev) for _ in range(2)]
        full = [torch.empty((world * n, R_), dtype=torch.uint8, device=dev) for _ in range(2)]
        host_full = torch.empty((world * n, R_), dtype=torch.uint8) if shared_gpu else None
        side = None if shared_gpu else torch.cuda.Stream(dev)
        n5 = c5["envs"]
        b5 = EnvBatch(n5, 30, 30, c5["max_trial"], "arc", dev)
        b5.set_op_table(actions.table_descs(ARCEnv.default_operations()))
        b5.set_tasks_padded(*make_tasks_c5(n5, 1000 + rank))
        b5.reset()
        FL5 = b5.elide_flag | c5["flags"]
        bb5_np, op5_np = make_actions_c5(S, n5, 2000 + rank)
        bb5, op5 = torch.from_numpy(bb5_np).to(dev), torch.from_numpy(op5_np).to(dev)
        torch.cuda.synchronize(dev)
    except Exception as exc:  # noqa: BLE001
        err = f"{type(exc).__name__}: {exc}"
    bad = max_over_ranks(0.0 if err is None else 1.0)
    if bad:
        return {"error": err or "setup failed on another rank"}

    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream

    def all_gather(slot):
        if shared_gpu:
            dist.all_gather_into_tensor(host_full, packed[slot].cpu())
        else:
            dist.all_gather_into_tensor(full[slot], packed[slot])

    def step4(i):
        b4.set_packed_output(packed[i & 1])  # (launch parameters are taken by value)
        b4.step_bbox_ptr(bb4[i % S].data_ptr(), op4[i % S].data_ptr(), FL4, sh)

    done = [None, None]

    def overlapped(i):
        slot = i & 1
        if done[slot] is not None:
            stream.wait_event(done[slot])  # the gather that last read this buffer
        step4(i)
        if side is None:
            all_gather(slot)
            return
        ev = torch.cuda.Event()
        ev.record(stream)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            all_gather(slot)
            done[slot] = torch.cuda.Event()
            done[slot].record(side)

    def serial(i):
        step4(i)
        all_gather(i & 1)

    def gather_only(i):
        all_gather(i & 1)

    def step_only(i):
        step4(i)

    def timed(fn):
        for i in range(4):  # warm-up (RCCL builds its channels on the first call)
            fn(i)
        if side is not None:
            stream.wait_stream(side)
        torch.cuda.synchronize(dev)
        ts = []
        for r in range(R):
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(K):
                fn(4 + i)
            if side is not None:
                stream.wait_stream(side)
            torch.cuda.synchronize(dev)
            dist.barrier()
            ts.append((time.perf_counter() - t0) / K)
        return max_over_ranks(float(np.median(ts)))

    t_over = timed(overlapped)
    done[0] = done[1] = None
    t_serial = timed(serial)
    t_gather = timed(gather_only)
    t_step = timed(step_only)
    assert b4.status() == 0
    shard_bytes = n * R_
    res["c4"] = {"workload": c4["name"], "envs_per_gpu": n, "global_envs": n * world, "row_bytes": R_,
                 "us_per_step": t_over * 1e6, "serial_us_per_step": t_serial * 1e6, "gather_only_us": t_gather * 1e6,
                 "step_only_us_eager": t_step * 1e6, "value": n * world / t_over, "unit": "env-steps/s",
                 "gather_bytes_per_rank_in": shard_bytes * (world - 1),
                 "gather_GBps": shard_bytes * world / t_gather / 1e9,
                 "gather_GBps_per_link_dir": shard_bytes / t_gather / 1e9,
                 "gather_note": "gather_GBps = gathered tensor bytes / collective time (algorithm bandwidth); per_link_dir = one shard / collective "
                                "time: what each directed xGMI link carries when every peer pair exchanges its shard directly",
                 "transport": "gloo through a host bounce (ranks share one GPU: functional, not a measurement)" if shared_gpu else "RCCL all_gather_into_tensor"}

    # ---- c5: sharded, no collective -------------------------------------------------------------------------------------------------
    def enqueue5(sh_):
        for i in range(K):
            b5.step_bbox_ptr(bb5[i].data_ptr(), op5[i].data_ptr(), FL5, sh_)
    for i in range(4):
        b5.step_bbox_ptr(bb5[K + i].data_ptr(), op5[K + i].data_ptr(), FL5, sh)
    dist.barrier()
    sec5, g5 = graph_time(dev, enqueue5, K, reps=max(R, 5))
    t5 = max_over_ranks(sec5)
    assert b5.status() == 0
    res["c5"] = {"workload": c5["name"], "envs_per_gpu": n5, "global_envs": n5 * world, "us_per_step": t5 * 1e6,
                 "value": n5 * world / t5, "unit": "env-steps/s", "collective": "none (envs are independent)"}
    del g5
    return res



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
    ap.add_argument("--action-batches", type=int, default=0, help="distinct action batches staged in HBM (default: up to 2048, the regions cycle "
                    "through them; a small number keeps the action stream cache-resident — a policy that writes one batch per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the non-headline legs (kernel A/B runs)")
    ap.add_argument("--no-multi", action="store_true", help="N > 1: skip the bounded c4 (step + RCCL all-gather) and c5 (sharded FloodFill) legs after the c3 headline")
    ap.add_argument("--no-forms", "--no-ordered", dest="no_ordered", action="store_true", help="skip the `forms` leg (the region's K launches as ONE arcle_step_many call, "
                    "and the K single-step calls with the dispatch order off)")
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
        import datetime
        # (a bounded wait: a rank that dies between two collectives must not leave the others in RCCL's default 10-minute watchdog — every
        # phase of an N > 1 run is seconds long; the CPU baseline and the extras legs run at N = 1 only)
        tmo = datetime.timedelta(seconds=int(os.environ.get("ARCLE_BENCH_DIST_TIMEOUT", "300")))
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)

    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv, O2ARCv2Env

    cfg = CONFIGS[a.config]
    H, W, kind = cfg["H"], cfg["W"], cfg["kind"]
    n = a.envs_per_gpu or cfg["envs"]
    K, Wm = a.steps, a.warmup
    R = a.regions or max(5, min(50, math.ceil(2000 / max(K, 1))))
    if shared_gpu and a.config == "c4" and not a.regions:
        R = 3  # (ranks sharing one GPU gather through a CPU bounce, ~0.2 s per step: a functional leg, not a measurement)
    S = min(K * R + Wm, max(K + Wm, 2048))  # distinct action batches staged in HBM (regions cycle through them)
    if a.action_batches:
        S = max(K + Wm, a.action_batches)

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
    side = None
    if a.config == "c4":  # the learner-side gather: ONE all-gather of the packed 912-byte record per env and step
        # The step kernel writes the packed rows itself (STEP_PACK_OBS, fused epilogue): no packing launch.  The rows are double-
        # buffered and the collective runs on a SIDE stream behind an event: step i+1 (writing the other buffer) overlaps the
        # all-gather of step i; a buffer is only rewritten once the gather that read it two steps ago has finished.  The sustained
        # rate is then max(step, collective) per step instead of their sum.
        R_ = batch.packed_obs_size()
        packed2 = [torch.zeros((n, R_), dtype=torch.uint8, device=dev) for _ in range(2)]
        batch.set_packed_output(packed2[0])
        FL |= STEP_PACK_OBS
        full2 = [torch.empty((world * n, R_), dtype=torch.uint8, device=dev) for _ in range(2)] if dist is not None else None
        host_full = torch.empty((world * n, R_), dtype=torch.uint8) if shared_gpu else None
        side = torch.cuda.Stream(dev) if dist is not None and not shared_gpu else None
        done = [None, None]

        def gather(slot, cur):
            if dist is None:
                return  # one rank, no process group: the packed rows ARE the gathered tensor
            if shared_gpu:
                dist.all_gather_into_tensor(host_full, packed2[slot].cpu())
                return
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dist.all_gather_into_tensor(full2[slot], packed2[slot])
                done[slot] = torch.cuda.Event()
                done[slot].record(side)

    def step(i, sh_=None, cur=None):
        j = i % S
        cur = cur if cur is not None else stream
        if gather is not None:
            slot = i & 1
            if done[slot] is not None:  # the gather that last read this buffer
                cur.wait_event(done[slot])
            batch.set_packed_output(packed2[slot])  # (launch parameters are taken by value: earlier launches keep their buffer)
        batch.step_bbox_ptr(bptr[j], optr[j], FL, sh if sh_ is None else sh_)
        if gather is not None:
            gather(i & 1, cur)

    def join_side(cur=None):
        if side is not None:
            (cur if cur is not None else stream).wait_stream(side)

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
    join_side()
    torch.cuda.synchronize(dev)
    # snapshot of the state region 0 starts from (replayed below for the byte accounting)
    snap = {k: v.clone() for k, v in batch.planes.items()}
    snap_rec, snap_cnt = batch.rec.clone(), batch.cnt.clone()

    # The K steps of a region are captured once into a hipGraph (K launches of arcle_step_kernel, each with its own action batch)
    # and replayed per region: a launch-bound inner loop belongs in a graph, and the host then issues one call per region instead
    # of K.  (With more than one rank c4 keeps eager launches when ranks share a GPU: its collective goes through the host.)
    # HEADLINE = the step()-per-call form: K arcle_step_bbox calls, none of which knows the next step's actions (the loop
    # a(t+1) = policy(obs(t)) of examples/example_bbox.py:13-15); every launch orders itself.  c3 also times, beside it
    # and with the same clock, the same K launches enqueued by ONE arcle_step_many call (what ARCVecEnv.capture records) and the K
    # single-step calls with the dispatch order off (the plain instantiation) -> the top-level `forms` block.
    graph = None
    graph_many = graph_plain = None
    many = a.config == "c3" and gather is None and not a.no_ordered and K > 1
    many_out = (torch.empty((K, n), dtype=torch.int32, device=dev), torch.empty((K, n), dtype=torch.uint8, device=dev)) if many else None
    if not a.no_graph and not (gather is not None and shared_gpu):
        try:
            graph = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(dev)
            cap.wait_stream(stream)
            if gather is not None:
                done[0] = done[1] = None
            with torch.cuda.graph(graph, stream=cap):
                cs = torch.cuda.current_stream(dev)
                for i in range(Wm, Wm + K):  # (c4 with a process group: the RCCL collectives are captured as well, on the forked side stream)
                    step(i, cs.cuda_stream, cs)
                join_side(cs)
            if gather is not None:
                done[0] = done[1] = None
            if many:
                graph_many = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_many, stream=cap):
                    batch.step_many("bbox", bbox[Wm:Wm + K], op[Wm:Wm + K], FL, many_out[0], many_out[1])
                # ... and the single-step calls again with the dispatch order off (launch parameters are baked in at capture)
                batch.set_dispatch_order(False)
                graph_plain = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_plain, stream=cap):
                    cs = torch.cuda.current_stream(dev)
                    for i in range(Wm, Wm + K):
                        step(i, cs.cuda_stream, cs)
                batch.set_dispatch_order(True)
        except Exception as exc:  # capture unsupported: eager launches
            print(f"bench: hipGraph capture failed ({exc}); eager launches", file=sys.stderr)
            graph = graph_many = graph_plain = None
            batch.set_dispatch_order(True)
            if gather is not None:
                done[0] = done[1] = None

    def region(r):
        if graph is not None:
            graph.replay()
        else:
            for i in range(Wm + r * K, Wm + (r + 1) * K):
                step(i)
            join_side()

    def restore_snapshot():
        for k, v in snap.items():  # back to the state the regions are defined to start from
            batch.planes[k].copy_(v)
        batch.rec.copy_(snap_rec)
        batch.cnt.copy_(snap_cnt)

    # untimed clock ramp (the chip idles at low clocks before the first launch): the region's own launches, ~80 ms of them
    t_ramp, r = time.perf_counter(), 0
    n_ramp = max(2, min(50, 4000 // max(K, 1)))  # (a fixed count when ranks must stay in step with each other)
    while not a.no_ramp and ((time.perf_counter() - t_ramp < 0.08) if dist is None else (r < n_ramp)):
        region(r)
        r += 1
        torch.cuda.synchronize(dev)
    restore_snapshot()
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
    def time_graph(g):  # same event clock and the same number of regions as the headline; the state simply keeps evolving
        for w_ in range(3):
            g.replay()
        ts = []
        for r in range(min(R, 15)):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            ev0.record(stream)
            g.replay()
            ev1.record(stream)
            wait_gpu(ev1)
            ts.append(ev0.elapsed_time(ev1) * 1e-3)
        return float(np.median(ts))
    forms = None
    if graph_many is not None:  # same event clock, this rank (N x this for the node)
        forms = {}
        for key, g_, what in (("step_many", graph_many, "ONE arcle_step_many call for the K steps (what ARCVecEnv.capture records)"),
                              ("dispatch_order_off", graph_plain, "K arcle_step_bbox calls after arcle_set_dispatch_order(env, 0): the plain instantiation, "
                                                                  "every wave steps the env of its own dispatch slot")):
            t_ = time_graph(g_)
            forms[key] = {"value": K * n * world / t_, "unit": "env-steps/s", "ms_per_step": t_ / K * 1e3, "avg_launch_us": t_ / K * 1e6, "form": what}
    wall_t = torch.tensor(devt if device_clock else wall, dtype=torch.float64)
    ranks_seen = world
    if dist is not None:  # max over ranks, per region
        wt = wall_t if shared_gpu else wall_t.to(dev)
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
        wall_t = wt.cpu()
        one = torch.ones(1, dtype=torch.float64) if shared_gpu else torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(one)  # every rank of the group adds 1: what the collective itself saw
        ranks_seen = int(one.item())
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = "?"
        print(f"bench: rank {rank}/{dist.get_world_size()} backend={dist.get_backend()} RCCL {rccl} device={dev} ({torch.cuda.get_device_name(dev)}) "
              f"envs [{rank * n}, {(rank + 1) * n}) median region {float(np.median(devt if device_clock else wall)) * 1e3:.4f} ms", file=sys.stderr, flush=True)
    elapsed = float(wall_t.median())
    kernel_avg_s = float(np.median(kern))
    status = batch.status()
    assert status == 0, f"device status {status}"
    total_steps = K * n * world

    # ---- bytes of the K launches of region 0, counted by the kernel itself: restore the snapshot and replay them (untimed) with
    #      the accounting instantiation — algorithmic bytes (the numerator of `frac`) and issued bytes (`traffic`) ----
    roofline = None
    if rank == 0:
        restore_snapshot()
        if gather is not None:
            batch.set_packed_output(packed2[0])

        def replay_region0(sh_):  # (the accounting instantiation hands out envs in index order: the bytes do not depend on the order)
            for i in range(Wm, Wm + K):
                batch.step_bbox_ptr(bptr[i % S], optr[i % S], FL, sh_)
        per_launch_bytes, issued_bytes, _ = counted_bytes(batch, replay_region0, K, dev)
        grp = "|grouped" if batch.orders_itself("bbox", FL) else ""
        kname = {"c3": f"arcle_step_kernel<bbox, FULL, 0, 0, autoreset|elide{grp}, 30>",
                 "c4": f"arcle_step_kernel<bbox, FULL, 0, 0, autoreset|elide|pack{grp}, 30> (fused packed-row epilogue; a multi-rank run has the all-gather inside the event pair)"}.get(a.config, "arcle_step_kernel")
        roofline = roofline_block(kname, kernel_avg_s, per_launch_bytes, issued_bytes, n, PS=batch.PS, planes=len(batch.planes),
                                  plan=batch.launch_info("bbox", FL),
                                  note="algorithmic bytes follow SURVEY.md 8d and include the reset_sel zero-fills of `selected` that "
                                       "ARCLE_STEP_ELIDE_SELECTED never writes (about 14 % of the figure on this mix); `traffic` does not")
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and a.config == "c3" and n == CONFIGS["c3"]["envs"]:
            pmc = json.load(open(pmc_path))
            roofline["pmc_crosscheck"] = {"hbm_bytes_per_launch": pmc.get("hbm_bytes_per_launch"),
                                          "source": "recorded, not measured in this run: " + str(pmc.get("source"))}
        if a.config == "c2":
            roofline.update({"bound": "launch-floor", "launch_floor_us": 2.5})
        if a.config == "c5":
            roofline.update({"bound": "iteration (dependent flood-fill passes)"})
        for f_ in (forms or {}).values():
            f_["frac"] = per_launch_bytes / (f_["avg_launch_us"] * 1e-6) / HBM_PEAK

    multi = None
    if dist is not None and world > 1 and a.config == "c3" and not a.no_multi:
        try:
            multi = multi_legs(dev, dist, world, rank, shared_gpu, K=4 if shared_gpu else 20, R=2 if shared_gpu else 3)
        except Exception as exc:  # noqa: BLE001 - (a local failure after the agreement point; the headline line is still printed)
            multi = {"error": f"{type(exc).__name__}: {exc}"}

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
                       else (f"env-shard x{world} + one packed all-gather per step ({'gloo, shared GPU' if shared_gpu else 'RCCL on a side stream, double-buffered rows, overlapping the next step'})"
                             if dist is not None else "one rank: step with the fused packed-row epilogue, nothing to gather")},
            "headline_form": "per_step_calls: K arcle_step_* calls, one launch each, none sees the next step's actions (every launch orders itself)",
            "timing": {"regions": R, "stat": "median region, max over ranks per region",
                       "clock": "HIP events on the launch stream, recorded between the region's two synchronisations" if device_clock else "host perf_counter between the region's two synchronisations",
                       "host_region_ms": [round(x * 1e3, 4) for x in wall][:12],
                       "launch": "hipGraph of the K step launches (K arcle_step_bbox calls), one replay per region" if graph is not None else "eager",
                       "region_ms": [round(float(x) * 1e3, 4) for x in wall_t.tolist()][:12]},
            "forms": forms,
            "roofline": roofline,
        }
        if dist is not None:
            out["collective"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks_seen": ranks_seen,
                                 "devices_visible": ndev, "shared_gpu": bool(shared_gpu),
                                 "data_path": "none (envs are independent)" if gather is None else "one all_gather_into_tensor of the packed rows per step"}
        if multi is not None:
            out["multi"] = multi
        if a.config == "c5":
            seeds = [(int(bbox_np[0, e, 0]), int(bbox_np[0, e, 1])) for e in range(n) if 10 <= op_np[0, e] < 20]
            grids = [tasks[0][e] for e in range(n) if 10 <= op_np[0, e] < 20]
            out["floodfill"] = {"share_of_actions": float(((op_np >= 10) & (op_np < 20)).mean()),
                                "frontier_rounds": frontier_rounds_sample(grids, seeds)}
        if world == 1 and not a.no_extras and a.config == "c3":
            ex = {}
            for name, leg in (("vec_api", lambda: vec_api_leg(dev, n, bbox, op)),
                              ("research_env", lambda: research_env_leg(dev, n, bbox, op)),
                              ("mask_ingress", lambda: ingress_leg(dev, n, bbox, op)),
                              ("host_actions", lambda: host_actions_leg(dev, n, bbox, op)),
                              ("single_env", lambda: single_env_leg(dev)),
                              ("transition_rows", lambda: transition_leg(dev, n, bbox, op)),
                              ("rollout", lambda: rollout_leg(batch, bbox, op, dev)),
                              ("batch_sweep", lambda: batch_sweep_leg(dev, bbox, op)),
                              ("other_configs", lambda: other_configs_leg(dev)), ("big_grid", lambda: big_grid_leg(dev))):
                try:
                    ex[name] = leg()
                except Exception as exc:  # an extra must never cost the headline line
                    ex[name] = {"error": f"{type(exc).__name__}: {exc}"}
                torch.cuda.empty_cache()
            out["extras"] = ex
        if world == 1 and not a.no_cpu_baseline and a.config == "c3":
            sustain = Sustained(graph, dev, K * n, K * kernel_avg_s * 1e6) if graph is not None else None
            out["cpu_baseline"] = cpu_baseline(1000, sustain)
            if sustain is not None:
                out["sustained"] = sustain.result()
        emit(out, a)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
