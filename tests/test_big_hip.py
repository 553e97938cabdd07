"""Grids beyond 1024 cells on the GPU: the workgroup-per-env kernels (arcle_amd/csrc/arcle_big.hip) through the C ABI, against the golden
vectors captured from the reference at those sizes and against the oracle.  The reference takes any max_grid_size (base.py:37-49)."""
import numpy as np
import pytest

import backends as B
import bigcases as C
from oracle import oracle as O
from oracle import refdriver as RD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", B.big_fixture_names())
def test_hip_reproduces_big_golden(name):
    errs = B.replay_fixture(B.HipBackend, name)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(40, 40), (33, 48), (48, 33), (64, 64), (127, 127), (127, 9), (9, 127), (100, 20)])
@pytest.mark.parametrize("kind,flags,max_trial", [("o2arc", 0, -1), ("o2arc", 3, 3), ("arc", 1, 3), ("raw", 0, 2)])
def test_hip_big_random_traces_vs_oracle(H, W, kind, flags, max_trial):
    O.set_threads(16)
    try:
        errs = B.random_trace_compare(B.HipBackend, kind, O.KIND_OPS[kind](), H, W, N=48, S=48, seed=H * 131 + W + flags, max_trial=max_trial,
                                      flags=flags, bad_ops=True)
    finally:
        O.set_threads(1)
    assert not errs, "\n".join(errs[:10])


def test_hip_big_batch_of_2048_envs_50x50():
    """every env of a 2048-env batch against the oracle (the workgroup count exceeds the chip's resident workgroups several times over)"""
    O.set_threads(16)
    try:
        errs = B.random_trace_compare(B.HipBackend, "o2arc", O.o2arc_ops(), 50, 50, N=2048, S=24, seed=77, max_trial=3, flags=3, new_forms=False)
    finally:
        O.set_threads(1)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(40, 40), (36, 41), (127, 127)])
def test_hip_big_record_and_bit_packed_forms(H, W):
    """5-tuple records (arcle_step_bbox5) and bit-packed boolean masks (arcle_step_bits, rows of plane_stride / 8 bytes) against the oracle's
    classic forms; arcle_pack_mask_bits"""
    errs = B.random_trace_compare(B.HipBackend, "o2arc", O.o2arc_ops(), H, W, N=24, S=60, seed=H + W, max_trial=3, flags=3, new_forms=True)
    assert not errs, "\n".join(errs[:10])
    rng = np.random.default_rng(1)
    be = B.HipBackend(24, H, W, 3, "o2arc", O.o2arc_ops())
    assert be.b.bits_stride == ((H * W + 127) & ~127) // 8
    m = (rng.random((24, H, W)) < 0.3).astype(np.int8) * rng.integers(-3, 4, (24, H, W)).astype(np.int8)
    assert np.array_equal(be.pack_mask_bits(m), B.pack_bits(m))


@pytest.mark.parametrize("H,W,flags", [(40, 40, 3), (33, 48, 3), (64, 64, 3), (100, 20, 3), (127, 127, 3), (127, 9, 3), (48, 40, 0)])
def test_hip_big_int8_masks_of_any_value(H, W, flags):
    """masks of arbitrary int8 values (negative, > 1, the {2, -1} pair whose sum is 1): the whole-word reductions of the mask ingest in the
    LEAN kernels (one wavefront per env at 40 x 40, eight at 127 x 127) and the per-cell form at W < 16 — every field of every env against
    the oracle"""
    w = [3] * 10 + [3] * 10 + [2] * 8 + [2] * 7
    errs = B.random_trace_compare(B.HipBackend, "o2arc", O.o2arc_ops(), H, W, N=48, S=40, seed=H + 3 * W + flags, flags=flags, max_trial=3, op_weights=w,
                                  int8_masks=True)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("variant", ["o2arc_exotic", "o2arc_crop"])
@pytest.mark.parametrize("size", [45, 127])
def test_hip_big_exotic_tables(variant, size):
    kind, ops = RD.variant_table(variant)
    w = [1] * 35
    for k in range(20, 28):
        w[k] = 4
    errs = B.random_trace_compare(B.HipBackend, kind, ops, size, size, N=32, S=100, seed=len(variant) + size, op_weights=w, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(40, 40), (64, 64), (127, 127), (33, 100)])
def test_hip_big_floodfill_worst_case(H, W):
    errs = B.floodfill_worst_case_compare(B.HipBackend, H, W)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("kind", ["o2arc", "arc", "raw"])
def test_hip_big_rows(kind):
    C.rows_case(B.HipBackend, kind)


def test_hip_big_truncation_and_autoreset():
    C.truncation_case(B.HipBackend)


def test_hip_big_task_table_and_device_draw():
    C.task_table_case(B.HipBackend)


@pytest.mark.parametrize("H,W", [(40, 40), (64, 64), (127, 127)])
def test_hip_big_dense_pair(H, W):
    """ARCLE_STEP_DENSE (agents/env.py:44-58 as an exact integer pair) incl. the (0, 0) of auto-reset and skipped steps"""
    import rows as R
    errs = R.dense_on_autoreset(B.HipBackend, H, W)
    assert not errs, "\n".join(errs[:10])


def test_hip_big_transition_rows():
    """the stateless batched transition (o2arcenv.py:149-151) on grids of more than 1024 cells: rows in, scratch envs, rows out"""
    import rows as R
    errs = R.transition_rows(B.HipBackend, cases=(("o2arc", 40, 40, 3), ("arc", 36, 41, 3), ("raw", 35, 30, 2), ("o2arc", 127, 127, -1)))
    assert not errs, "\n".join(errs[:10])


def test_hip_big_byte_accounting():
    """arcle_enable_accounting on a big handle: CopyFromInput of inactive envs (zero-fill of `selected` elided) moves two planes per env"""
    H, W, N = 40, 40, 64
    be = B.HipBackend(N, H, W, 3, "o2arc", O.o2arc_ops())
    inp = np.random.default_rng(0).integers(0, 10, (N, H, W)).astype(np.int8)
    dims = np.full((N, 2), H, np.int8)
    be.set_tasks(inp, dims, inp, dims)
    be.reset()
    be.b.enable_accounting(True)
    be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 31, np.int32), 2)
    alg, issued, steps = be.b.accounting_ex(clear=True)
    scal = 2 * 16 + 16 + 20 + 5
    assert (alg, issued, steps) == (N * (2 * H * W + scal), N * (2 * be.b.PS + scal), N)
    be.b.enable_accounting(False)


def test_hip_big_task_augmentation():
    C.aug_case(B.HipBackend)


def test_hip_big_research_env_flags():
    """ARCVecEnv as the research env (dense reward, TimeLimit, device-drawn tasks with augmentation at every auto-reset, FilterO2ARC rows) at
    max_grid_size (40, 40): runs, stays consistent with its own stand-alone row writer, reports no device error"""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    loader = SyntheticLoader(n_tasks=8, seed=3, max_size=(40, 40))
    venv = ARCVecEnv(O2ARCv2Env, 64, loader, max_grid_size=(40, 40), autoreset="resample", seed=5, max_episode_steps=6, dense_reward=True,
                     augment=True)
    rows = venv.enable_flat_rows(filtered=True)
    obs, info = venv.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(20):
        bbox = torch.randint(0, 40, (64, 4), generator=g, dtype=torch.int32).cuda()
        op = torch.randint(0, 35, (64,), generator=g, dtype=torch.int32).cuda()
        obs, reward, term, trunc, info = venv.step_bbox(bbox, op)
        torch.cuda.synchronize()
        assert reward.dtype == torch.float32 and int(info["steps"].max()) <= 6
        assert torch.equal(rows, venv.flat_obs(filtered=True))
    venv.check_errors()
    assert int(venv.batch.episode.max()) >= 2  # envs restarted on freshly drawn tasks


def test_hip_big_rollout_is_n_step_launches():
    """arcle_rollout_bbox on a big handle = n_steps step launches (the state does not fit a wavefront): same results as stepping"""
    H = W = 40
    N, T = 24, 12
    rng = np.random.default_rng(8)
    a = B.HipBackend(N, H, W, 3, "o2arc", O.o2arc_ops())
    b_ = B.HipBackend(N, H, W, 3, "o2arc", O.o2arc_ops())
    inp = rng.integers(0, 10, (N, H, W)).astype(np.int8)
    dims = np.full((N, 2), H, np.int8)
    for be in (a, b_):
        be.set_tasks(inp, dims, inp, dims)
        be.reset()
    bbox = np.stack([rng.integers(0, H, (T, N)), rng.integers(0, W, (T, N)), rng.integers(0, H, (T, N)), rng.integers(0, W, (T, N))], 2)
    op = rng.integers(0, 35, (T, N))
    r, t = a.rollout("bbox", bbox, op, 3)
    for s in range(T):
        r1, t1 = b_.step("bbox", bbox[s], op[s], 3)
        assert np.array_equal(r[s], r1) and np.array_equal(t[s], t1)
    for f in ("grid", "selected", "object", "clip"):
        assert np.array_equal(a.get(f), b_.get(f))


def test_hip_big_vec_env_and_single_env_api():
    """ARCVecEnv and the Gymnasium single-env class at max_grid_size (40, 40): reset, steps, observations against the oracle"""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    loader = SyntheticLoader(n_tasks=8, seed=3, max_size=(40, 40))
    venv = ARCVecEnv(O2ARCv2Env, 64, loader, max_grid_size=(40, 40), autoreset=True, rng=np.random.default_rng(0), seed=5)
    obs, info = venv.reset()
    assert obs["grid"].shape == (64, 40, 40)
    orc = B.OracleBackend(64, 40, 40, -1, "o2arc", O.o2arc_ops())
    orc.set_tasks(venv.batch.plane("input").cpu().numpy(), venv.batch.field("input_dim").cpu().numpy(),
                  venv.batch.plane("answer").cpu().numpy(), venv.batch.field("answer_dim").cpu().numpy())
    orc.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    for _ in range(16):
        bbox = torch.randint(0, 40, (64, 4), generator=g, dtype=torch.int32)
        op = torch.randint(0, 35, (64,), generator=g, dtype=torch.int32)
        obs, reward, term, trunc, info = venv.step_bbox(bbox.cuda(), op.cuda())
        r2, t2 = orc.step("bbox", bbox.numpy(), op.numpy(), 1)
        assert np.array_equal(reward.cpu().numpy(), r2) and np.array_equal(term.cpu().numpy().astype(np.uint8), t2)
        assert np.array_equal(obs["grid"].cpu().numpy(), orc.get("grid"))
        assert np.array_equal(obs["selected"].cpu().numpy(), orc.get("selected"))
    venv.check_errors()
    env = O2ARCv2Env(data_loader=loader, max_grid_size=(40, 40), colors=10)
    obs, info = env.reset(options={"prob_index": 1, "subprob_index": 0})
    o1 = B.OracleBackend(1, 40, 40, -1, "o2arc", O.o2arc_ops())
    inp = np.zeros((1, 40, 40), np.int8)
    ans = np.zeros((1, 40, 40), np.int8)
    inp[0, :env.input_.shape[0], :env.input_.shape[1]] = env.input_
    ans[0, :env.answer.shape[0], :env.answer.shape[1]] = env.answer
    o1.set_tasks(inp, np.array([env.input_.shape], np.int8), ans, np.array([env.answer.shape], np.int8))
    o1.reset()
    rng = np.random.default_rng(2)
    for _ in range(20):
        sel = np.zeros((40, 40), np.int8)
        x, y = rng.integers(0, 36), rng.integers(0, 36)
        sel[x:x + rng.integers(1, 5), y:y + rng.integers(1, 5)] = 1
        op = int(rng.integers(0, 35))
        obs, reward, term, trunc, info = env.step({"selection": sel, "operation": op})
        r2, t2 = o1.step("mask", sel[None], np.array([op], np.int32), 0)
        assert reward == int(r2[0]) and term == bool(t2[0])
        assert np.array_equal(obs["grid"], o1.get("grid")[0]) and np.array_equal(obs["selected"], o1.get("selected")[0])
        assert np.array_equal(obs["object_states"]["object"], o1.get("object")[0])
        assert info["steps"] == int(o1.counters()[0, 0])
    # transition(state, action) (o2arcenv.py:149-151; README: env.transition(deepcopy(state), action)): the env itself stays where it is
    import copy
    st = copy.deepcopy(env.current_state)
    steps_before = env.action_steps
    sel = np.zeros((40, 40), np.int8)
    sel[2:9, 3:11] = 1
    env.transition(st, {"selection": sel, "operation": 22})  # MoveR of a fresh selection
    r2, t2 = o1.step("mask", sel[None], np.array([22], np.int32), 0)
    assert np.array_equal(st["grid"], o1.get("grid")[0]) and np.array_equal(st["selected"], o1.get("selected")[0])
    assert np.array_equal(st["object_states"]["object_pos"], o1.get("object_pos")[0]) and env.action_steps == steps_before
    assert not np.array_equal(env.current_state["grid"], st["grid"]) or not sel.any()
