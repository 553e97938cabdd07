"""GPU parity tests proper: the HIP path (arcle_amd -> libarcle_hip.so C ABI -> gfx950 kernels) against
  (1) the committed golden vectors captured from the imported reference,
  (2) the oracle on seeded random traces over many grid sizes / env kinds / ingress forms,
  (3) at BASELINE.json's full size (8192 envs, 30x30): exact parity on a sample of envs, independence of a
      env's trajectory from the batch it is stepped in, determinism, and conservation properties.
Bit-exact everywhere: the path is int8/byte arithmetic."""
import numpy as np
import pytest

import backends as B
from oracle import oracle as O

pytestmark = pytest.mark.gpu

OBJ_HEAVY = [1] * 10 + [2] * 10 + [4] * 8 + [2] * 3 + [1] * 4


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from arcle_amd import _lib
    _lib.build()  # no-op when csrc/libarcle_hip.so is up to date (it travels with the snapshot)
    _lib.lib()    # the product library must be present and loadable: no silent fallback


@pytest.mark.parametrize("name", B.fixture_names())
def test_hip_matches_golden(name):
    errs = B.replay_fixture(B.HipBackend, name)
    assert not errs, "\n".join(errs[:10])
    fx = B.load_fixture(name)
    if fx["meta"]["kind"] == "o2arc" and B.can_elide(fx["meta"]["ops"]):
        # the same vectors with ARCLE_STEP_ELIDE_SELECTED (what ARCVecEnv and bench.py pass): identical states
        errs = B.replay_fixture(B.HipBackend, name, flags=B.STEP_ELIDE_SELECTED)
        assert not errs, "elide: " + "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(30, 30), (10, 10), (5, 5), (3, 3), (7, 12), (12, 7), (32, 32), (1, 17), (20, 1),
                                 (16, 16), (2, 100), (15, 17)])
def test_hip_vs_oracle_o2arc(H, W):
    for flags in (0, O.STEP_AUTORESET, B.STEP_ELIDE_SELECTED, O.STEP_AUTORESET | B.STEP_ELIDE_SELECTED):
        errs = B.random_trace_compare(B.HipBackend, "o2arc", O.o2arc_ops(), H, W, N=96, S=96, seed=H * 100 + W + flags,
                                      max_trial=3 if flags else -1, flags=flags, op_weights=OBJ_HEAVY, bad_ops=True)
        assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("kind,ops", [("arc", O.arc_ops()), ("raw", O.raw_ops())])
@pytest.mark.parametrize("H,W", [(30, 30), (10, 10), (5, 5)])
def test_hip_vs_oracle_other_kinds(kind, ops, H, W):
    errs = B.random_trace_compare(B.HipBackend, kind, ops, H, W, N=64, S=96, seed=11 + H, max_trial=3)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(30, 30), (16, 16), (20, 17), (17, 20), (12, 12), (9, 32), (32, 32)])
def test_hip_vs_oracle_exotic_ops(H, W):
    """Rot180, Flip D0/D1, keep_sel, paste_blank=False, Crop ... (tables no shipped env installs)."""
    from oracle import refdriver as RD
    w = [1] * 20 + [4] * 8 + [2] * 7
    errs = B.random_trace_compare(B.HipBackend, "o2arc", RD.variant_table("o2arc_exotic")[1], H, W, N=96, S=96,
                                  seed=3 * H + W, op_weights=w)
    assert not errs, "\n".join(errs[:10])


def test_hip_floodfill_stress():
    """Config-5 style: ARCEnv table, 70 % FloodFill point seeds on few-colour grids (long frontiers)."""
    ops = O.arc_ops()
    w = [1] * 10 + [7] * 10 + [1] * 7
    errs = B.random_trace_compare(B.HipBackend, "arc", ops, 30, 30, N=128, S=64, seed=5, max_trial=3, op_weights=w)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(30, 30), (32, 32), (17, 21), (12, 12), (6, 40), (16, 16), (31, 33)])
def test_hip_floodfill_worst_case(H, W):
    """Spiral corridor (graph diameter ~H*W/2) and full-board regions: the frontier loop must converge exactly."""
    errs = B.floodfill_worst_case_compare(B.HipBackend, H, W)
    assert not errs, "\n".join(errs)


def _full_size_run(N, S, seed, sample):
    """Runs N envs for S bbox steps on the GPU; returns the final state of the `sample` envs + everything needed
    to re-run just those envs elsewhere."""
    import torch
    from arcle_amd.engine import EnvBatch
    rng = np.random.default_rng(seed)
    H = W = 30
    inp = np.zeros((N, H, W), np.int8)
    ans = np.zeros((N, H, W), np.int8)
    idim = rng.integers(1, 31, (N, 2)).astype(np.int8)
    adim = idim.copy()
    full = rng.integers(0, 10, (N, H, W)).astype(np.int8)
    rows, cols = np.arange(H)[None, :, None], np.arange(W)[None, None, :]
    inside = (rows < idim[:, 0, None, None]) & (cols < idim[:, 1, None, None])
    inp[inside] = full[inside]
    ans[:] = inp
    bbox = rng.integers(0, 30, (S, N, 4)).astype(np.int32)
    op = rng.integers(0, 35, (S, N)).astype(np.int32)
    b = EnvBatch(N, H, W, -1, "o2arc")
    b.set_op_table(O.o2arc_ops())
    b.set_tasks_padded(inp, idim, ans, adim)
    b.reset()
    bb, oo = torch.from_numpy(bbox).cuda(), torch.from_numpy(op).cuda()
    rewards = np.zeros((S, N), np.int32)
    for s in range(S):
        r, t = b.step_bbox(bb[s], oo[s], b.elide_flag)  # the flags bench.py / ARCVecEnv use
        rewards[s] = r.cpu().numpy()
    torch.cuda.synchronize()
    assert b.status() == 0
    state = {k: b.plane(k)[sample].cpu().numpy() for k in b.planes}
    state["rec"] = b.rec[sample].cpu().numpy()
    state["cnt"] = b.cnt[sample].cpu().numpy()
    state["grid_sum_all"] = int(b.plane("grid").to(torch.int64).sum())
    return state, (inp, idim, ans, adim, bbox, op, rewards)


def test_full_size_8192_sample_parity_and_batch_independence():
    """BASELINE config 3 shape: 8192 envs, 30x30, all 35 ops uniform, BBox tuples uniform."""
    N, S = 8192, 64
    sample = np.r_[0:64, 4064:4128, 8128:8192, np.random.default_rng(1).choice(N, 192, replace=False)]
    got, (inp, idim, ans, adim, bbox, op, rewards) = _full_size_run(N, S, seed=77, sample=sample)
    # (a) exact parity of the sampled envs against the oracle stepping ONLY those envs
    orc = B.OracleBackend(len(sample), 30, 30, -1, "o2arc", O.o2arc_ops())
    orc.set_tasks(inp[sample], idim[sample], ans[sample], adim[sample])
    orc.reset()
    for s in range(S):
        r, _ = orc.step("bbox", bbox[s][sample], op[s][sample])
        assert np.array_equal(r, rewards[s][sample]), f"reward mismatch at step {s}"
    for k in O.KIND_PLANES["o2arc"]:
        assert np.array_equal(got[k], orc.env.planes[k]), f"plane {k} differs at full batch size"
    assert np.array_equal(got["rec"], orc.env.rec) and np.array_equal(got["cnt"], orc.env.cnt)
    # (b) determinism: the same run again gives the same device state
    again, _ = _full_size_run(N, S, seed=77, sample=sample)
    for k in got:
        assert np.array_equal(np.asarray(got[k]), np.asarray(again[k])), f"{k} not deterministic"


def test_properties_at_full_size():
    """Size-independent properties on 8192 envs: CopyFromInput is idempotent and restores grid==input;
    ResetGrid zeroes; Color with the full-grid bbox then Submit of a matching answer terminates with reward 1;
    a Move followed by the opposite Move (object fully inside) restores the grid."""
    import torch
    from arcle_amd.engine import EnvBatch
    N, H, W = 8192, 30, 30
    rng = np.random.default_rng(3)
    inp = rng.integers(1, 10, (N, H, W)).astype(np.int8)
    dims = np.full((N, 2), 30, np.int8)
    ans = np.full((N, H, W), 7, np.int8)
    b = EnvBatch(N, H, W, -1, "o2arc")
    b.set_op_table(O.o2arc_ops())
    b.set_tasks_padded(inp, dims, ans, dims)
    b.reset()
    dev = b.device
    ones = lambda v: torch.full((N,), v, dtype=torch.int32, device=dev)  # noqa: E731
    box = lambda x1, y1, x2, y2: torch.tensor([[x1, y1, x2, y2]] * N, dtype=torch.int32, device=dev)  # noqa: E731
    tin = torch.from_numpy(inp).to(dev)
    # Move right then left of an interior block restores the grid; `selected` ends where it started
    b.step_bbox(box(5, 5, 9, 12), ones(22))
    b.step_bbox(box(0, 0, 0, 0) * 0 + torch.tensor([40, 40, 41, 41], dtype=torch.int32, device=dev), ones(23))  # empty selection: continue
    assert torch.equal(b.plane("grid"), tin)
    sel = b.plane("selected")
    assert int(sel.sum()) == N * 5 * 8 and bool((sel[:, 5:10, 5:13] == 1).all())
    # ResetGrid / CopyFromInput
    b.step_bbox(box(0, 0, 0, 0), ones(32))
    assert int(b.plane("grid").abs().sum()) == 0
    b.step_bbox(box(0, 0, 0, 0), ones(31))
    b.step_bbox(box(0, 0, 0, 0), ones(31))
    assert torch.equal(b.plane("grid"), tin)
    # Color7 everywhere + Submit -> terminated, reward 1, submit_count 1
    b.step_bbox(box(0, 0, 29, 29), ones(7))
    r, t = b.step_bbox(box(0, 0, 0, 0), ones(34))
    assert int(r.sum()) == N and int(t.sum()) == N and int(b.cnt[:, 1].sum()) == N
    assert int(b.cnt[:, 0].min()) == 7 and int(b.cnt[:, 0].max()) == 7
    assert b.status() == 0


@pytest.mark.parametrize("H,W", [(30, 30), (10, 10), (5, 7)])
def test_hip_reset_from_task_table(H, W):
    errs = B.task_table_compare(B.HipBackend, H, W, N=200, T=37, seed=H)
    assert not errs, "\n".join(errs)


def test_vec_env_reset_and_resample_autoreset():
    """ARCVecEnv.reset goes through the device task table; autoreset='resample' restarts a finished env on a new task at
    its next step (Gymnasium next-step autoreset, inside the step kernel)."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    loader = SyntheticLoader(n_tasks=11, max_size=(8, 8), seed=3, p_same=1.0)
    venv = ARCVecEnv(O2ARCv2Env, 512, loader, max_grid_size=(8, 8), max_trial=2, autoreset="resample",
                     rng=np.random.default_rng(0))
    obs, info = venv.reset(options={"adaptation": True})
    # every env holds the (input, answer) pair its task_index / subprob_index names
    for n in (0, 17, 511):
        t, s_ = int(info["task_index"][n]), int(info["subprob_index"][n])
        a = loader.data[t][0][s_]
        assert tuple(obs["input_dim"][n].tolist()) == a.shape
        assert np.array_equal(obs["input"][n, :a.shape[0], :a.shape[1]].cpu().numpy(), a)
        assert np.array_equal(obs["grid"][n].cpu().numpy(), obs["input"][n].cpu().numpy())
    # submit immediately: answer == input (p_same=1) -> every env terminates with reward 1 ...
    op = torch.full((512,), 34, dtype=torch.int32, device=venv.device)
    box = torch.zeros((512, 4), dtype=torch.int32, device=venv.device)
    first = info["table_index"].clone()
    obs, reward, term, trunc, info = venv.step_bbox(box, op)
    assert int(reward.sum()) == 512 and bool(term.all()) and int(obs["terminated"].sum()) == 512
    # ... and starts a fresh episode on a newly drawn task at its next step (that step's action is not executed)
    obs, reward, term, trunc, info = venv.step_bbox(box, op)
    assert int(reward.sum()) == 0 and not bool(term.any())
    assert int(obs["terminated"].sum()) == 0 and int(info["steps"].max()) == 0 and int(obs["trials_remain"].min()) == 2
    assert int((info["table_index"] != first).sum()) > 256 and int(venv.batch.episode.min()) == 2
    venv.check_errors()
    # indexed reset of a masked subset
    mask = torch.zeros(512, dtype=torch.bool)
    mask[:100] = True
    obs, info = venv.reset(options={"prob_index": 4, "subprob_index": 0, "adaptation": False}, env_mask=mask)
    a = loader.data[4][2][0]
    assert np.array_equal(obs["input"][5, :a.shape[0], :a.shape[1]].cpu().numpy(), a)
    assert (info["task_index"][:100] == 4).all()
    with pytest.raises(AssertionError):
        venv.reset(options={"prob_index": 99})


@pytest.mark.parametrize("H,W,ingress", [(30, 30, "bbox"), (10, 10, "bbox"), (30, 30, "point"), (5, 7, "bbox"), (16, 24, "bbox")])
def test_hip_rollout_equals_sequential_steps(H, W, ingress):
    """arcle_rollout_* (T steps in one launch, planes resident in registers) == T oracle steps."""
    for flags in (0, O.STEP_AUTORESET, B.STEP_ELIDE_SELECTED):
        errs = B.rollout_compare(B.HipBackend, "o2arc", O.o2arc_ops(), H, W, N=160, T=64, seed=H + W + flags,
                                 ingress=ingress, flags=flags)
        assert not errs, "\n".join(errs[:10])
    for kind, ops in (("arc", O.arc_ops()), ("raw", O.raw_ops())):
        errs = B.rollout_compare(B.HipBackend, kind, ops, H, W, N=64, T=48, seed=3, ingress=ingress)
        assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("kind,ops", [("arc", O.arc_ops()), ("raw", O.raw_ops())])
def test_flattened_observation_other_kinds(kind, ops):
    """Env kinds without some keys omit them from the flattened row (the O2ARCv2Env layouts are pinned on the reference by
    tests/features.py::flat)."""
    H = W = 30
    be = B.HipBackend(16, H, W, 3, kind, ops)
    rng = np.random.default_rng(1)
    inp = rng.integers(0, 10, (16, H, W)).astype(np.int8)
    dims = np.tile(np.array([[H, W]], np.int8), (16, 1))
    be.set_tasks(inp, dims, inp, dims)
    be.reset()
    be.step("bbox", rng.integers(0, 30, (16, 4)).astype(np.int32), rng.integers(0, len(ops), 16).astype(np.int32))
    keys = (["clip", "clip_dim"] if kind == "arc" else []) + ["grid", "grid_dim", "input", "input_dim", "terminated", "trials_remain"]
    want = np.concatenate([be.get(k).reshape(16, -1) for k in keys], 1)
    assert np.array_equal(be.flat_obs(), want)


def test_single_env_gym_api_matches_oracle():
    """The Gymnasium-style single env (reference API: dict obs, dict action) on the GPU."""
    from arcle_amd.envs import O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    from arcle_amd.wrappers import BBoxWrapper
    loader = SyntheticLoader(n_tasks=3, max_size=(10, 10), seed=5)
    env = O2ARCv2Env(data_loader=loader, max_grid_size=(10, 10), colors=10, max_trial=3)
    obs, info = env.reset(options={"prob_index": 1, "subprob_index": 0})
    ti, to = loader.data[1][0][0], loader.data[1][1][0]
    orc = O.OracleEnv(1, 10, 10, 3, "o2arc")
    orc.set_tasks([ti], [to])
    orc.reset()
    assert set(obs) == {"trials_remain", "terminated", "input", "input_dim", "grid", "grid_dim", "selected", "clip",
                        "clip_dim", "object_states"}
    assert info["input_dim"] == ti.shape and info["steps"] == 0 and info["submit_count"] == 0
    wenv = BBoxWrapper(env)
    rng = np.random.default_rng(0)
    for _ in range(60):
        a = (int(rng.integers(0, 10)), int(rng.integers(0, 10)), int(rng.integers(0, 10)), int(rng.integers(0, 10)),
             int(rng.integers(0, 35)))
        obs, reward, term, trunc, info = wenv.step(a)
        r2, t2 = orc.step_bbox([a[:4]], [a[4]])
        ref = orc.state_dict(0)
        assert reward == int(r2[0]) and term == bool(t2[0]) and trunc is False
        for k, v in ref.items():
            if k == "object_states":
                for k2, v2 in v.items():
                    assert obs[k][k2].dtype == np.int8 and np.array_equal(obs[k][k2], v2), k2
            else:
                assert obs[k].dtype == np.int8 and np.array_equal(obs[k], v), k
        assert info["steps"] == orc.cnt[0, 0] and info["submit_count"] == orc.cnt[0, 1]
    with pytest.raises(IndexError):
        env.step({"selection": np.zeros((10, 10), np.int8), "operation": 35})
    # transition(state, action) mutates the given dict in place and leaves the env untouched
    import copy
    st = copy.deepcopy(obs)
    steps_before = env.action_steps
    env.transition(st, {"selection": np.ones((10, 10), np.int8), "operation": 4})
    assert (st["grid"] == 4).all() and env.action_steps == steps_before


def test_custom_operation_table_like_reference_subclasses():
    """create_operations() overrides as in agents/env.py:23-28 (op 33 := reset_sel(crop_grid)) and
    agents/wrapper.py:53-57 (dropping ops) map onto device tables."""
    from arcle_amd import actions as A
    from arcle_amd.envs import O2ARCv2Env, ARCVecEnv
    from arcle_amd.loaders import SyntheticLoader

    class CustomEnv(O2ARCv2Env):
        def create_operations(self):
            ops = super().create_operations()
            ops[33] = A.reset_sel(A.crop_grid)
            return ops

    env = CustomEnv(data_loader=SyntheticLoader(n_tasks=2, max_size=(8, 8), min_size=(8, 8), seed=1), max_grid_size=(8, 8))
    assert env.op_names[33] == "CropGrid"
    obs, _ = env.reset(options={"prob_index": 0, "subprob_index": 0})
    g0 = obs["grid"].copy()
    sel = np.zeros((8, 8), np.int8)
    sel[2:5, 1:4] = 1
    obs, *_ = env.step({"selection": sel, "operation": 33})
    assert obs["grid_dim"].tolist() == [3, 3] and np.array_equal(obs["grid"][:3, :3], g0[2:5, 1:4])
    # (arbitrary Python callables in the table run on the host: tests/test_features_hip.py::test_host_callable_in_the_op_table)
    nofill = O2ARCv2Env.default_operations()
    nofill = nofill[:10] + nofill[20:]
    venv = ARCVecEnv(O2ARCv2Env, 32, SyntheticLoader(n_tasks=4, seed=2), operations=nofill)
    assert len(venv.op_names) == 25 and venv.op_names[-1] == "Submit"


def test_hip_config5_workload_vs_oracle():
    """bench.py's c5 workload itself (ARCEnv 27-op table, stripes / blobs / spiral grids, 70 % FloodFill from in-bounds point
    seeds): every env of a 1024-env batch equals the oracle after every step — the row-board flood fill under load, mixed with
    the other ops of the table."""
    import bench
    N, S = 1024, 24
    ops = O.arc_ops()
    inp, dims, ans, adims = bench.make_tasks_c5(N, 11)
    bbox, op = bench.make_actions_c5(S, N, 13)
    be, orc = B.HipBackend(N, 30, 30, -1, "arc", ops), B.OracleBackend(N, 30, 30, -1, "arc", ops)
    for b_ in (be, orc):
        b_.set_tasks(inp, dims, ans, adims)
        b_.reset()
    for s in range(S):
        r1, t1 = be.step("bbox", bbox[s], op[s])
        r2, t2 = orc.step("bbox", bbox[s], op[s])
        assert np.array_equal(r1, r2) and np.array_equal(t1, t2), s
        for f in ("grid", "grid_dim", "clip", "clip_dim"):
            assert np.array_equal(be.get(f), orc.get(f)), (s, f)
    assert be.status() == orc.status() == 0
