"""Round-4 GPU tests: the oracle meets exactly what bench.py times, at the sizes BASELINE.json names.

  * the headline kernel `arcle_step_kernel<bbox, FULL, 0, 0, autoreset|elide, 30>` (K single-step calls) and the `ordered` leg's
    `<..., autoreset|elide|ordered, 30>` (one arcle_step_many call), 8192 envs x 64 steps of bench.py's own task / action streams,
    against the oracle stepping a sample of the envs (envs are independent, o2arcenv.py:130-151, so a sample is exact);
  * c4's per-node batch on ONE GPU: 65 536 envs stepped with the fused packed-row epilogue, the packed rows unpacked and compared;
  * c5 at 32 768 envs (ARCEnv, 70 % flood fills);
  * a 300-step slice of tools/soak.py;
  * `bench.py --gpus 2` on one GPU (the N > 1 path: spawn, rendezvous, sharding, max-over-ranks, the compact last line).
Bit-exact everywhere."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import backends as B
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS_O2 = ["grid", "selected", "clip", "object", "object_sel", "background", "input", "answer", "input_dim", "grid_dim", "clip_dim", "object_dim",
             "object_pos", "trials_remain", "terminated", "active", "rotation_parity", "answer_dim"]
FIELDS_ARC = ["grid", "clip", "input", "answer", "input_dim", "grid_dim", "clip_dim", "trials_remain", "terminated", "answer_dim"]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from arcle_amd import _lib
    _lib.build()
    _lib.lib()


def _sample(n, k, seed):
    edge = np.r_[0:32, n // 2 - 16:n // 2 + 16, n - 32:n]
    return np.unique(np.r_[edge, np.random.default_rng(seed).choice(n, k, replace=False)])


def _oracle_for(sample, tasks, H, W, max_trial, kind, ops):
    orc = B.OracleBackend(len(sample), H, W, max_trial, kind, ops)
    orc.set_tasks(*(t[sample] for t in tasks))
    orc.reset()
    return orc


def _compare_sample(batch, orc, sample, fields, what):
    import torch
    idx = torch.as_tensor(sample, device=batch.device)
    for f in fields:
        got = (batch.plane(f) if f in batch.planes else batch.field(f))[idx].cpu().numpy()
        assert np.array_equal(got, orc.get(f)), f"{what}: field {f} differs from the oracle"
    assert np.array_equal(batch.cnt[idx].cpu().numpy(), orc.counters()), f"{what}: counters differ"


@pytest.mark.parametrize("form", ["per_step_calls", "ordered_step_many", "ordered_bbox5"])
@pytest.mark.parametrize("max_trial", [-1, 3])
def test_bench_headline_kernels_vs_oracle_8192(form, max_trial):
    """What bench.py times, checked DIRECTLY against the oracle: 8192 envs x 64 steps of bench.make_tasks / make_actions with
    ARCLE_STEP_AUTORESET | ARCLE_STEP_ELIDE_SELECTED — as K arcle_step_bbox calls (`value`) and as one arcle_step_many call with
    ordered dispatch (`ordered`; bbox + op arrays and 5-tuple records).  Sample: 448 envs, every field, reward and terminated of
    every step; the sticky status must stay 0."""
    import torch
    import bench
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import O2ARCv2Env
    n, K, dev = 8192, 64, torch.device("cuda:0")
    tasks = bench.make_tasks(n, 1000)
    bb_np, op_np = bench.make_actions(K, n, 2000)
    batch = EnvBatch(n, 30, 30, max_trial, "o2arc", dev)
    batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    batch.set_tasks_padded(*tasks)
    batch.reset()
    FL = batch.elide_flag | 1
    assert FL == 3, "the O2ARC table permits the zero-fill elision: this is the instantiation bench.py launches"
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    if form == "per_step_calls":
        rew, trm = torch.empty((K, n), dtype=torch.int32, device=dev), torch.empty((K, n), dtype=torch.uint8, device=dev)
        for s in range(K):
            r, t = batch.step_bbox(bb[s], op[s], FL)
            rew[s], trm[s] = r, t
    elif form == "ordered_step_many":
        batch.set_dispatch_order(True)
        rew, trm = batch.step_many("bbox", bb, op, FL)
    else:
        batch.set_dispatch_order(True)
        rew, trm = batch.step_many("bbox5", torch.cat([bb, op[:, :, None]], 2).contiguous(), None, FL)
    torch.cuda.synchronize()
    assert batch.status() == 0
    sample = _sample(n, 352, 5)
    orc = _oracle_for(sample, tasks, 30, 30, max_trial, "o2arc", O.o2arc_ops())
    rew, trm = rew.cpu().numpy(), trm.cpu().numpy()
    ended = 0
    for s in range(K):
        r2, t2 = orc.step("bbox", bb_np[s][sample], op_np[s][sample], O.STEP_AUTORESET)
        assert np.array_equal(rew[s][sample], r2) and np.array_equal(trm[s][sample], t2), f"{form}: reward / terminated differ at step {s}"
        ended += int(t2.sum())
    assert ended > 0, "the trace must exercise the auto-reset path"
    _compare_sample(batch, orc, sample, FIELDS_O2, form)


def test_c4_65536_envs_packed_rows_vs_oracle():
    """BASELINE configs[3]'s per-node batch on one GPU: 65 536 envs (state 0.5 GB, the streaming regime: 4-wave workgroups) stepped
    with the fused packed-row epilogue ShardedVecEnv uses; after every step the packed rows of the sampled envs — grid | grid_dim |
    reward | terminated — and at the end every state field equal the oracle's."""
    import torch
    import bench
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch, STEP_PACK_OBS
    from arcle_amd.envs import O2ARCv2Env
    n, K, dev = 65536, 20, torch.device("cuda:0")
    tasks = bench.make_tasks(n, 1000)
    bb_np, op_np = bench.make_actions(K, n, 2000)
    batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
    batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    batch.set_tasks_padded(*tasks)
    batch.reset()
    packed = batch.set_packed_output()
    packed.fill_(0x55)
    FL = batch.elide_flag | 1 | STEP_PACK_OBS
    sample = _sample(n, 416, 6)
    idx = torch.as_tensor(sample, device=dev)
    orc = _oracle_for(sample, tasks, 30, 30, -1, "o2arc", O.o2arc_ops())
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    for s in range(K):
        r, t = batch.step_bbox(bb[s], op[s], FL)
        r2, t2 = orc.step("bbox", bb_np[s][sample], op_np[s][sample], O.STEP_AUTORESET)
        grid, gdim, rew, term = EnvBatch.unpack_obs(packed[idx], 30, 30)
        assert np.array_equal(rew.cpu().numpy(), r2) and np.array_equal(term.cpu().numpy(), t2.astype(bool)), s
        assert np.array_equal(r[idx].cpu().numpy(), r2) and np.array_equal(t[idx].cpu().numpy(), t2), s
        assert np.array_equal(grid.cpu().numpy(), orc.get("grid")) and np.array_equal(gdim.cpu().numpy(), orc.get("grid_dim")), s
        assert not packed[idx][:, 907:].any(), "row padding must be zero"
    assert batch.status() == 0
    _compare_sample(batch, orc, sample, FIELDS_O2, "c4 @ 65536")
    # every row of the batch, not only the sample: the packed grid IS the grid plane, reward / terminated ARE the step outputs
    grid, gdim, rew, term = EnvBatch.unpack_obs(packed, 30, 30)
    assert torch.equal(grid, batch.plane("grid")) and torch.equal(gdim, batch.field("grid_dim"))
    assert torch.equal(rew, batch.reward) and torch.equal(term, batch.term != 0)


def test_c5_32768_envs_floodfill_vs_oracle():
    """BASELINE configs[4] at its full size on one GPU: ARCEnv 27-op table, 32 768 envs, 70 % FloodFill point seeds on stripes / blobs /
    spiral corridors (bench.make_tasks_c5 / make_actions_c5), 12 steps; a 1056-env sample against the oracle, every field."""
    import torch
    import bench
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv
    n, K, dev = 32768, 12, torch.device("cuda:0")
    tasks = bench.make_tasks_c5(n, 1000)
    bb_np, op_np = bench.make_actions_c5(K, n, 2000)
    batch = EnvBatch(n, 30, 30, -1, "arc", dev)
    batch.set_op_table(actions.table_descs(ARCEnv.default_operations()))
    batch.set_tasks_padded(*tasks)
    batch.reset()
    FL = batch.elide_flag | 1
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    rew, trm = batch.step_many("bbox", bb, op, FL)
    torch.cuda.synchronize()
    assert batch.status() == 0
    sample = _sample(n, 960, 7)
    O.set_threads(8)
    try:
        orc = _oracle_for(sample, tasks, 30, 30, -1, "arc", O.arc_ops())
        rew, trm = rew.cpu().numpy(), trm.cpu().numpy()
        for s in range(K):
            r2, t2 = orc.step("bbox", bb_np[s][sample], op_np[s][sample], O.STEP_AUTORESET)
            assert np.array_equal(rew[s][sample], r2) and np.array_equal(trm[s][sample], t2), s
    finally:
        O.set_threads(1)
    _compare_sample(batch, orc, sample, FIELDS_ARC, "c5 @ 32768")
    filled = (batch.plane("grid") != batch.plane("input")).flatten(1).any(1).float().mean().item()
    assert filled > 0.5, "most envs must have seen an effective flood fill"


def test_soak_slice():
    """300 steps of tools/soak.py (every state field after every step vs the oracle: the lean 30x30 instantiations with runtime and
    compile-time flags, FAST / GENERIC width classes, ARCEnv flood fills, 5-tuple / bit-packed ingress, the research flag set with
    the dense cache and incremental rows)."""
    env = dict(os.environ, SOAK_STEPS="300")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py")], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if " S=300" in ln]
    assert len(lines) >= 11, p.stdout
    bad = [ln for ln in lines if ": OK" not in ln]
    assert not bad, "\n".join(bad)


def _bench(args, launcher=None, timeout=900):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last) < 4000, f"the last stdout line must stay compact ({len(last)} bytes)"
    return json.loads(last), p


@pytest.mark.parametrize("cfg", ["c3", "c4"])
def test_bench_two_ranks_on_one_gpu(cfg):
    """The N > 1 path without a multi-GPU node: `python bench.py --gpus 2` spawns two ranks that share cuda:0 (gloo control plane),
    shards the global batch, takes the max over ranks per region and prints ONE compact, parseable last line."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("the shared-GPU leg is for one-GPU boxes")
    out, p = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", cfg, "--regions", "3", "--no-cpu-baseline", "--no-extras"])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_envs"] == 2 * out["config"]["envs_per_gpu"] == 16384
    c = out["collective"]
    assert c["world"] == 2 and c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["shared_gpu"] is True
    assert out["value"] > 0 and abs(out["value"] - 6 * 16384 / (out["ms_per_step"] * 6e-3)) / out["value"] < 1e-3
    assert out["roofline"]["frac"] > 0 and out["roofline"]["algorithmic_bytes_per_launch"] > 8192 * 1000
    ranks = [ln for ln in p.stderr.splitlines() if ln.startswith("bench: rank ")]
    assert len(ranks) == 2 and any("rank 0/2" in ln for ln in ranks) and any("rank 1/2" in ln for ln in ranks), p.stderr[-1500:]


def test_bench_one_rank_is_the_same_line_with_and_without_a_launcher():
    """SCALE(N = 1) is BENCH: `python bench.py --gpus 1` and the driver's torchrun form run the same code path and print the same keys
    (no process group, no `collective` block), and the stdout of either ends in the compact line."""
    args = ["--gpus", "1", "--steps", "8", "--warmup", "2", "--regions", "3", "--no-cpu-baseline", "--no-extras"]
    plain, _ = _bench(args)
    launched, _ = _bench(args, launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                                         "--master-port", "29533"])
    assert sorted(plain) == sorted(launched) and "collective" not in plain
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "config", "headline_form", "dtype", "scaling"):
        assert plain[k] == launched[k], k
    assert plain["roofline"]["algorithmic_bytes_per_launch"] == launched["roofline"]["algorithmic_bytes_per_launch"]
    assert plain["ordered"]["value"] > 0 and plain["roofline"]["kernel"].startswith("arcle_step_kernel<bbox, FULL, 0, 0, autoreset|elide, 30>")
