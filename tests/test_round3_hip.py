"""GPU suite, round 3: the paths users call (ARCVecEnv multi-step / captured stepping, the batched stateless transition, state
checkpoints, the single-env class's one-launch step, dense reward across auto-resets, augmentation with caller-chosen tasks), the
overlapped gather over RCCL, the by-value ordering of the ABI setters, and BASELINE's c2 / c5 workloads at their real shapes —
all against the oracle or against the equivalent sequence of plain steps."""
import copy
import os

import numpy as np
import pytest

import backends as B
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


def _loader(n_tasks=12, seed=4, size=(12, 12)):
    from arcle_amd.loaders import SyntheticLoader
    return SyntheticLoader(n_tasks=n_tasks, seed=seed, max_size=size)


def _actions(K, N, H, W, seed, submit=0.15):
    import torch
    g = torch.Generator().manual_seed(seed)
    bb = torch.stack([torch.randint(0, H, (K, N), generator=g), torch.randint(0, W, (K, N), generator=g),
                      torch.randint(0, H, (K, N), generator=g), torch.randint(0, W, (K, N), generator=g)], -1).int().cuda()
    op = torch.where(torch.rand((K, N), generator=g) < submit, torch.tensor(34), torch.randint(0, 35, (K, N), generator=g)).int().cuda()
    return bb.contiguous(), op.contiguous()


@pytest.mark.parametrize("kw", [dict(autoreset=True), dict(autoreset="resample", max_episode_steps=7, dense_reward=True, augment=True),
                                dict(max_trial=2)], ids=["autoreset", "research", "plain"])
def test_step_many_and_capture_equal_single_steps(kw):
    """ARCVecEnv.step_many(K actions) and ARCVecEnv.capture(...).replay() == K calls of step_bbox: per-step reward / terminated /
    truncated and the final state, incl. dense rewards (0 on auto-reset steps), TimeLimit and device-drawn new tasks."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    N, K = 96, 24
    mk = lambda: ARCVecEnv(O2ARCv2Env, N, _loader(), max_grid_size=(12, 12), seed=11, **kw)  # noqa: E731
    a, b, c = mk(), mk(), mk()
    for v in (a, b, c):
        v.reset()
    bb, op = _actions(K, N, 12, 12, 3)
    rs, ts, trs = [], [], []
    for i in range(K):
        o, r, t, tr, info = a.step_bbox(bb[i], op[i])
        rs.append(r.clone()), ts.append(t.clone()), trs.append(tr.clone())
    o2, r2, t2, tr2, info2 = b.step_many(bb, op)
    cs = c.capture(bb, op)
    o3, r3, t3, tr3 = cs.replay()
    for name, (rr, tt, ttr, vv) in (("step_many", (r2, t2, tr2, b)), ("capture", (r3, t3, tr3, c))):
        assert torch.equal(torch.stack(rs), rr), name
        assert torch.equal(torch.stack(ts), tt) and torch.equal(torch.stack(trs), ttr), name
        for k in a.batch.planes:
            assert torch.equal(a.batch.planes[k], vv.batch.planes[k]), (name, k)
        assert torch.equal(a.batch.rec, vv.batch.rec) and torch.equal(a.batch.cnt, vv.batch.cnt), name
    if kw.get("dense_reward"):
        assert r2.dtype == torch.float32 and bool((torch.stack(ts)[:-1].float().sum() > 0))  # episodes really ended
        ended = torch.stack(ts)[:-1] | torch.stack(trs)[:-1]
        assert bool((torch.stack(rs)[1:][ended] == 0).all()), "the auto-reset step of an env must carry reward 0"
    # a second replay continues from the new state with the SAME captured action buffers rewritten in place
    bb2, op2 = _actions(K, N, 12, 12, 4)
    cs.payload.copy_(bb2), cs.operation.copy_(op2)
    o3, r3, t3, tr3 = cs.replay()
    o2, r2, t2, tr2, _ = b.step_many(bb2, op2)
    assert torch.equal(r2, r3) and torch.equal(t2, t3) and torch.equal(b.batch.planes["grid"], c.batch.planes["grid"])
    for v in (a, b, c):
        v.check_errors()


def test_vec_transition_4096_states_from_the_golden_fixture():
    """ARCVecEnv.transition over 4096 states taken from the o2arc_30 golden trace x random actions == the oracle's step of the same
    (state, action) pairs; nothing of the env's own state moves; rows chain (transition of a transition)."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    fx = B.load_fixture("o2arc_30")
    m = fx["meta"]
    N0 = m["N"]
    orc = B.OracleBackend(N0, 30, 30, m["max_trial"], "o2arc", m["ops"])
    orc.set_tasks(fx["input"], fx["input_dim"], fx["answer"], fx["answer_dim"])
    orc.reset()
    mask_idx = {int(s): i for i, s in enumerate(fx["mask_steps"])}
    pool = []
    for s in range(min(m["S"], 64)):
        ing = int(fx["ingress"][s])
        if ing == 0:
            orc.step("bbox", fx["bbox"][s], fx["op"][s])
        elif ing == 1:
            orc.step("point", fx["xy"][s], fx["op"][s])
        else:
            orc.step("mask", fx["masks"][mask_idx[s]], fx["op"][s])
        pool.append((B.state_rows(orc), np.arange(N0)))
    orc.status()
    rows = np.concatenate([p[0] for p in pool])
    src = np.concatenate([p[1] for p in pool])
    rng = np.random.default_rng(8)
    pick = rng.integers(0, len(rows), 4096)
    rows, src = rows[pick], src[pick].astype(np.int32)
    M = len(rows)

    class _L:  # the fixture's tasks as a loader: task n = env n's pair
        data = [([fx["input"][n][:fx["input_dim"][n, 0], :fx["input_dim"][n, 1]]], [fx["answer"][n][:fx["answer_dim"][n, 0], :fx["answer_dim"][n, 1]]], [], [], {}) for n in range(N0)]
    v = ARCVecEnv(O2ARCv2Env, N0, _L(), max_grid_size=(30, 30), max_trial=m["max_trial"], seed=1)
    v.reset(options={"prob_index": np.arange(N0), "subprob_index": 0})
    before = v.state_rows().clone()
    big = B.OracleBackend(M, 30, 30, m["max_trial"], "o2arc", m["ops"])
    big.set_tasks(fx["input"][src], fx["input_dim"][src], fx["answer"][src], fx["answer_dim"][src])
    big.reset()
    lay, off = B.row_layout("o2arc", 900), 0
    for f, n in lay:
        dst = big.env.planes[f] if f in big.env.planes else big.env.field(f)
        dst[:] = rows[:, off:off + n].reshape(dst.shape)
        off += n
    cur = torch.from_numpy(rows).cuda()
    for rnd, form in enumerate(("bbox", "mask", "point")):
        ing, pay, op = R._random_actions(rng, M, 30, 30, 35)
        while ing != form:
            ing, pay, op = R._random_actions(rng, M, 30, 30, 35)
        act = {"operation": torch.from_numpy(op).cuda(), {"bbox": "bbox", "point": "point", "mask": "selection"}[form]: torch.from_numpy(pay).cuda()}
        out, rw, tm = v.transition(cur, act, src_env=torch.from_numpy(src).cuda())
        r2, t2 = big.step(form, pay, op)
        assert np.array_equal(out.cpu().numpy(), B.state_rows(big)), f"round {rnd} ({form}): rows differ"
        assert np.array_equal(rw.cpu().numpy(), r2) and np.array_equal(tm.cpu().numpy().astype(np.uint8), t2)
        big.status(), v.batch.status()
        cur = out.contiguous()
    assert torch.equal(v.state_rows(), before), "transition touched the env's own state"


def test_checkpoint_restore_continues_bit_identically():
    """get_state() / set_state(): a restored batch replays the same future (states, rewards, device-drawn tasks)."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    N, K = 128, 20
    v = ARCVecEnv(O2ARCv2Env, N, _loader(), max_grid_size=(12, 12), seed=21, autoreset="resample", max_episode_steps=6, augment=("rot90",))
    v.reset()
    bb, op = _actions(2 * K, N, 12, 12, 9)
    v.step_many(bb[:K], op[:K])
    ck = v.get_state()
    _, r1, t1, tr1, _ = v.step_many(bb[K:], op[K:])
    end1 = v.state_rows().clone()
    ep1 = v.batch.episode.clone()
    v.set_state(ck)
    _, r2, t2, tr2, _ = v.step_many(bb[K:], op[K:])
    assert torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(tr1, tr2)
    assert torch.equal(end1, v.state_rows()) and torch.equal(ep1, v.batch.episode)
    # rows alone restore the state dict (task side untouched): set_state_rows(state_rows()) is the identity
    rows = v.state_rows().clone()
    v.step_many(bb[:3], op[:3])
    v.set_state_rows(rows)
    assert torch.equal(v.state_rows(), rows)


def test_reset_with_chosen_tasks_still_augments():
    """augment=... with prob_index / subprob_index — the research env's own usage (agents/env.py:31-42 sets the task and
    augments on every reset).  The augmentation is what sampling.draw_aug_batch predicts for (seed, global env id, episode)."""
    import torch
    from arcle_amd import sampling
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    N = 40
    ld = _loader()
    v = ARCVecEnv(O2ARCv2Env, N, ld, max_grid_size=(12, 12), seed=5, env_base=100, augment=("permute", "rot90"))
    for episode in range(2):
        obs, info = v.reset(options={"prob_index": np.arange(N) % len(ld.data), "subprob_index": 0})
        k, perm = sampling.draw_aug_batch(5, 100 + np.arange(N), np.full(N, episode), 3)
        for n in range(N):
            g = np.asarray(ld.data[n % len(ld.data)][0][0])
            want = np.rot90(perm[n][g], int(k[n]))
            h, w = want.shape
            assert tuple(obs["input_dim"][n].tolist()) == (h, w), (episode, n)
            assert np.array_equal(obs["input"][n, :h, :w].cpu().numpy(), want), (episode, n)
        assert int(v.batch.episode.min()) == episode + 1
    assert len({tuple(p) for p in perm.tolist()}) > 1 and len(set(k.tolist())) > 1
    v.check_errors()


def test_single_env_step_is_one_launch_and_transition_is_stateless():
    """The single-env class: step() parity with the oracle through the pinned-row path (state, reward, terminated, counters,
    raised errors), and transition(deepcopy(state), action) == the oracle's step while the env's own state stays put."""
    from arcle_amd.envs import O2ARCv2Env
    ld = _loader(size=(30, 30))
    env = O2ARCv2Env(data_loader=ld, max_grid_size=(30, 30), max_trial=3)
    obs, info = env.reset(options={"prob_index": 2, "subprob_index": 0})
    orc = B.OracleBackend(1, 30, 30, 3, "o2arc", O.o2arc_ops())
    orc.env.set_tasks([env.input_], [env.answer])
    orc.reset()
    rng = np.random.default_rng(3)
    for s in range(60):
        ing, pay, op = R._random_actions(rng, 1, 30, 30, 35)
        if ing == "bbox":
            x1, y1, x2, y2 = pay[0]
            sel = np.zeros((30, 30), np.int8)
            sel[min(x1, x2):max(x1, x2) + 1, min(y1, y2):max(y1, y2) + 1] = 1
        elif ing == "point":
            sel = np.zeros((30, 30), np.int8)
            sel[pay[0, 0], pay[0, 1]] = 1
        else:
            sel = pay[0]
        action = {"selection": sel, "operation": int(op[0])}
        if s % 3 == 0:  # a look-ahead on a copy first: must not disturb the env
            st = copy.deepcopy(obs)
            shadow = B.OracleBackend(1, 30, 30, 3, "o2arc", O.o2arc_ops())
            shadow.env.set_tasks([env.input_], [env.answer])
            shadow.reset()
            for f in R._state_fields("o2arc"):
                (shadow.env.planes[f] if f in shadow.env.planes else shadow.env.field(f))[:] = orc.get(f)
            env.transition(st, action)
            shadow.step("mask", sel[None], op)
            assert np.array_equal(st["grid"], shadow.get("grid")[0]) and np.array_equal(st["selected"], shadow.get("selected")[0])
            assert np.array_equal(st["object_states"]["object"], shadow.get("object")[0])
            assert int(st["terminated"][0]) == int(shadow.get("terminated")[0, 0])
            shadow.status()
        obs, reward, term, trunc, info = env.step(action)
        r2, t2 = orc.step("mask", sel[None], op)
        assert reward == int(r2[0]) and term == bool(t2[0]), s
        for k in ("grid", "selected", "clip"):
            assert np.array_equal(obs[k], orc.get(k)[0]), (s, k)
        for k in ("object", "object_sel", "background"):
            assert np.array_equal(obs["object_states"][k], orc.get(k)[0]), (s, k)
        assert info["steps"] == int(orc.counters()[0, 0])
        orc.status()
        if term:
            break
    with pytest.raises(IndexError):
        env.step({"selection": np.zeros((30, 30), np.int8), "operation": 35})


def test_bbox_example_call_pattern():
    """The call pattern of the reference's examples/example_bbox.py against arcle_amd: make env, BBoxWrapper, reset, sample 5-tuples,
    step, reset on termination, close — every step checked against the oracle."""
    from arcle_amd.envs import O2ARCv2Env
    from arcle_amd.wrappers import BBoxWrapper
    env = O2ARCv2Env(data_loader=_loader(size=(30, 30)))  # = gym.make('ARCLE/O2ARCEnv') when gymnasium is installed
    env = BBoxWrapper(env)
    env.action_space[4].seed(0)
    for sp in env.action_space.spaces:
        sp.seed(1)
    obs, info = env.reset()
    action = env.action_space.sample()
    assert len(action) == 5
    base = env.unwrapped
    orc = B.OracleBackend(1, 30, 30, -1, "o2arc", O.o2arc_ops())

    def sync_oracle():
        orc.env.set_tasks([base.input_], [base.answer])
        orc.reset()
    sync_oracle()
    for _ in range(120):
        action = env.action_space.sample()
        obs, reward, term, trunc, info = env.step(action)
        r2, t2 = orc.step("bbox", np.asarray(action[:4], np.int32)[None], np.asarray([action[4]], np.int32))
        assert (reward, term) == (int(r2[0]), bool(t2[0])) and np.array_equal(obs["grid"], orc.get("grid")[0])
        if term or trunc:
            obs, info = env.reset()
            sync_oracle()
    env.close()


@pytest.mark.parametrize("cfg", ["c2", "c5"])
def test_baseline_workloads_at_their_real_shapes(cfg):
    """BASELINE configs c2 (O2ARCv2Env 10x10, 1024 envs, bench.make_actions_c2: 50 % rectangle / 40 % point / 10 % empty) and c5
    (ARCEnv 27 ops 30x30, 4096 envs, 70 % flood fills on stripes / blobs / spirals): the HIP path vs the oracle, every field."""
    import bench
    c = bench.CONFIGS[cfg]
    H, W, n, kind = c["H"], c["W"], c["envs"], c["kind"]
    K = 24
    if cfg == "c2":
        tasks, (bb, op), ops = bench.make_tasks(n, 1000, H, W, lo=3, zero_frac=0.5), bench.make_actions_c2(K, n, 2000), O.o2arc_ops()
    else:
        tasks, (bb, op), ops = bench.make_tasks_c5(n, 1000, H, W), bench.make_actions_c5(K, n, 2000, H, W), O.arc_ops()
    O.set_threads(8)
    try:
        be, orc = B.HipBackend(n, H, W, -1, kind, ops), B.OracleBackend(n, H, W, -1, kind, ops)
        for b in (be, orc):
            b.set_tasks(*tasks)
            b.reset()
        FL = 1 | (2 if B.can_elide(ops) and kind == "o2arc" else 0)
        for s in range(K):
            r1, t1 = be.step("bbox", bb[s], op[s], FL)
            r2, t2 = orc.step("bbox", bb[s], op[s], 1)
            assert np.array_equal(r1, r2) and np.array_equal(t1, t2), s
        assert be.status() == orc.status()
        for f in R._state_fields(kind):
            assert np.array_equal(be.get(f), orc.get(f)), f
        assert np.array_equal(be.counters(), orc.counters())
    finally:
        O.set_threads(1)


def test_setters_do_not_disturb_launches_in_flight():
    """ABI ordering: every launch (and every captured graph) holds its parameters by value.  Replacing the op table and the task
    table / sampler while a long graph replay is in flight changes nothing for it; launches enqueued afterwards see the new ones."""
    import torch
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch, STEP_RESAMPLE, STEP_TRUNCATE
    from arcle_amd.envs import O2ARCv2Env
    n, K = 4096, 200
    rng = np.random.default_rng(0)
    tasks_a = [(rng.integers(0, 10, (9, 9)).astype(np.int8),) * 2 for _ in range(8)]
    tasks_b = [(rng.integers(0, 10, (5, 7)).astype(np.int8),) * 2 for _ in range(8)]
    ops_a = actions.table_descs(O2ARCv2Env.default_operations())
    ops_b = list(reversed(ops_a[:-1])) + [ops_a[-1]]  # same ops, different slots
    bb, op = _actions(K, n, 12, 12, 5)

    def run(swap):
        b = EnvBatch(n, 12, 12, -1, "o2arc", "cuda")
        b.set_op_table(ops_a)
        b.set_task_table([t[0] for t in tasks_a], [t[1] for t in tasks_a])
        b.set_sampler(np.arange(8), np.ones(8), 3)
        b.set_truncation(5)
        b.reset_sampled()
        FL = b.elide_flag | STEP_RESAMPLE | STEP_TRUNCATE
        st = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            sh = torch.cuda.current_stream().cuda_stream
            for i in range(K):
                b.step_bbox_ptr(bb[i].data_ptr(), op[i].data_ptr(), FL, sh)
        g.replay()
        keep = (b._table, b._pair)  # (caller-owned arrays must outlive the launches that were given them)
        if swap:  # while the replay is (most likely) still running
            b.set_op_table(ops_b)
            b.set_task_table([t[0] for t in tasks_b], [t[1] for t in tasks_b])
            b.set_sampler(np.arange(8), np.ones(8), 99)
        torch.cuda.synchronize()
        del keep
        return b, {k: v.clone() for k, v in b.planes.items()}, b.rec.clone(), FL
    b0, p0, r0, FL = run(False)
    b1, p1, r1, _ = run(True)
    for k in p0:
        assert torch.equal(p0[k], p1[k]), f"a setter disturbed launches already enqueued ({k})"
    assert torch.equal(r0, r1)
    # launches enqueued AFTER the swap use the new tables: every env that restarts now gets a 5 x 7 task
    for i in range(12):
        b1.step_bbox(bb[i], op[i], FL)
    torch.cuda.synchronize()
    dims = b1.field("input_dim").cpu().numpy()
    assert ((dims == (5, 7)).all(1)).sum() > n // 2 and b1.status() == 0


def test_overlapped_gather_over_rccl_world_size_1():
    """ShardedVecEnv with the collective forced on (one rank, nccl = RCCL): gather_async on the side stream overlapping the next
    step, ping-pong groups, every = K windows and the hipGraph capture of step + all_gather all reproduce the local tensors."""
    import torch
    import torch.distributed as dist
    from arcle_amd.dist import ShardedVecEnv
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        G, K = 512, 12
        fac = lambda n, lo, hi: ARCVecEnv(O2ARCv2Env, n, _loader(), max_grid_size=(12, 12), seed=5, env_base=lo, autoreset=True)  # noqa: E731
        bb, op = _actions(K, G, 12, 12, 6)
        ref = fac(G, 0, G)
        ref.reset()
        want = []
        for i in range(K):
            o, r, t, tr, info = ref.step_bbox(bb[i], op[i])
            want.append((o["grid"].clone(), o["grid_dim"].clone(), r.clone(), t.clone()))
        # overlapped: wait for step i's rows only after step i+1 has been enqueued
        env = ShardedVecEnv(G, fac, force_collective=True)
        env.reset()
        pending, got = None, []
        for i in range(K):
            env.step_bbox(env.local_slice(bb[i]), env.local_slice(op[i]))
            w = env.gather_async()
            if pending is not None:
                got.append([x.clone() for x in pending.wait()])
                env.release(pending)
            pending = w
        got.append([x.clone() for x in pending.wait()])
        for i in range(K):
            assert all(torch.equal(a, b) for a, b in zip(got[i], want[i])), f"overlapped gather, step {i}"
        # ping-pong groups
        env = ShardedVecEnv(G, fac, groups=2, force_collective=True)
        env.reset()
        for i in range(K):
            ws = []
            for g in range(2):
                env.step_bbox(env.local_slice(bb[i], g), env.local_slice(op[i], g), group=g)
                ws.append(env.gather_async(g))
            for g in range(2):
                ids = env.group_global_ids(g)
                grid, gdim, r, t = ws[g].wait()
                assert torch.equal(grid, want[i][0][ids]) and torch.equal(r, want[i][2][ids]) and torch.equal(t, want[i][3][ids]), (i, g)
        # every = 4
        env = ShardedVecEnv(G, fac, every=4, force_collective=True)
        env.reset()
        for i in range(K):
            env.step_bbox(env.local_slice(bb[i]), env.local_slice(op[i]))
            if env.ready():
                grid, gdim, r, t = env.gather()
                for j in range(4):
                    assert torch.equal(grid[j], want[i - 3 + j][0]) and torch.equal(r[j], want[i - 3 + j][2]), (i, j)
        # step + collective captured in one hipGraph
        env = ShardedVecEnv(G, fac, force_collective=True)
        env.reset()
        cap = env.capture(bb, op)
        grid, gdim, r, t = cap.replay()
        torch.cuda.synchronize()
        for i in range(K):
            assert torch.equal(grid[i], want[i][0]) and torch.equal(gdim[i], want[i][1]) and torch.equal(r[i], want[i][2]) and torch.equal(t[i], want[i][3]), i
    finally:
        dist.destroy_process_group()


def test_kernel_counted_bytes():
    """arcle_get_accounting_ex: the algorithmic figure and the issued figure for a launch whose access pattern is
    known exactly — Color on every env of a 30 x 30 batch through bbox tuples."""
    import torch
    from arcle_amd import actions
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import O2ARCv2Env
    N = 64
    b = EnvBatch(N, 30, 30, -1, "o2arc", "cuda")
    b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    b.reset()
    b.enable_accounting(True)
    b.accounting_ex(clear=True)
    bb = torch.tensor([[2, 3, 9, 11]] * N, dtype=torch.int32, device="cuda")
    op = torch.full((N,), 4, dtype=torch.int32, device="cuda")
    b.step_bbox(bb, op, 0)             # Color: grid read + written, `selected` zero-filled (reset_sel)
    alg, issued, steps = b.accounting_ex(clear=True)
    assert steps == N and alg == N * (3 * 900 + 56)
    assert issued == N * (3 * 1024 + 32 + 16 + 20 + 5)
    b.step_bbox(bb, op, b.elide_flag)  # the same with the redundant zero-fill elided: one plane store less is issued
    alg2, issued2, _ = b.accounting_ex(clear=True)
    assert alg2 == alg and issued2 == issued - N * 1024


def test_host_slot_is_not_applied_to_envs_the_kernel_auto_reset():
    """with autoreset, the action of an env whose episode had ended is not executed (the launch re-initialises it) — a
    host-applied table slot must not touch that env either; for the others the callable's effect on `terminated` is reported."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader

    def paint(state, action):
        state["grid"][0, 0] = 7
        state["terminated"][0] = 1

    class Custom(O2ARCv2Env):
        def create_operations(self):
            ops = super().create_operations()
            ops[5] = paint
            return ops
    N = 8
    ld = SyntheticLoader(n_tasks=4, seed=2, max_size=(12, 12), min_size=(3, 3), p_same=1.0)
    v = ARCVecEnv(Custom, N, ld, max_grid_size=(12, 12), seed=1, autoreset=True)
    obs, info = v.reset()
    zeros = torch.zeros((N, 4), dtype=torch.int32, device="cuda")
    submit = torch.tensor([34, 34, 34, 34, 0, 0, 0, 0], dtype=torch.int32, device="cuda")
    obs, r, t, tr, info = v.step_bbox(zeros, submit)          # answer == input: envs 0-3 terminate with reward 1
    assert t.tolist() == [True] * 4 + [False] * 4 and r.tolist() == [1] * 4 + [0] * 4
    first = obs["input"][:, 0, 0].clone()
    obs, r, t, tr, info = v.step_bbox(zeros, torch.full((N,), 5, dtype=torch.int32, device="cuda"))
    g00 = obs["grid"][:, 0, 0]
    assert torch.equal(g00[:4], first[:4]), "the callable ran on envs that were auto-reset in this step"
    assert g00[4:].tolist() == [7] * 4 and t.tolist() == [False] * 4 + [True] * 4
    assert info["steps"].tolist() == [0] * 4 + [2] * 4
    v.check_errors()


def test_step_many_over_host_resident_records_with_in_launch_prefetch():
    """arcle_step_many on PINNED HOST 5-tuple records (30 x 30, ARCVecEnv's flags): step t reads what the front workgroups of launch
    t-1 copied into the device staging buffer — results equal the same steps fed from device memory, also when replayed in a graph."""
    import torch
    import bench
    n, K = 2048, 40
    bb, op = bench.make_actions(K, n, 77)
    act5 = torch.from_numpy(np.concatenate([bb, op[..., None]], -1).astype(np.int32))
    h5, d5 = act5.pin_memory(), act5.cuda()
    a, b = bench.make_batch(torch.device("cuda:0"), n, seed=5), bench.make_batch(torch.device("cuda:0"), n, seed=5)
    FL = a.elide_flag | 1
    ra, ta = a.step_many("bbox5", d5, None, FL)
    rb, tb = torch.empty_like(ra), torch.empty_like(ta)
    assert b.L.arcle_step_many(b._h, 3, K, h5.data_ptr(), None, rb.data_ptr(), tb.data_ptr(), FL, b._stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(ra, rb) and torch.equal(ta, tb)
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), k
    assert torch.equal(a.rec, b.rec) and torch.equal(a.cnt, b.cnt) and a.status() == 0 and b.status() == 0
    # the same inside a hipGraph, replayed twice with the host array rewritten in between
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=st):
        assert b.L.arcle_step_many(b._h, 3, K, h5.data_ptr(), None, rb.data_ptr(), tb.data_ptr(), FL, torch.cuda.current_stream().cuda_stream) == 0
    for seed in (78, 79):
        bb, op = bench.make_actions(K, n, seed)
        act5 = torch.from_numpy(np.concatenate([bb, op[..., None]], -1).astype(np.int32))
        torch.cuda.synchronize()
        h5.copy_(act5)
        g.replay()
        ra, ta = a.step_many("bbox5", act5.cuda(), None, FL)
        torch.cuda.synchronize()
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(a.planes["grid"], b.planes["grid"]) and torch.equal(a.rec, b.rec)


def test_plane_copies_and_step_many_in_every_action_form():
    """arcle_get_plane / arcle_set_plane (dense [N, H, W] <-> the strided plane, device and pinned host) and arcle_step_many for the
    mask / bit-packed / point forms: K steps per call == K single steps."""
    import torch
    from arcle_amd.engine import EnvBatch
    N, H, W, K = 64, 12, 12, 10
    rng = np.random.default_rng(4)
    mk = lambda: B.HipBackend(N, H, W, 2, "o2arc", O.o2arc_ops())  # noqa: E731
    a, b = mk(), mk()
    tasks = R._tasks(rng, N, H, W)
    for be in (a, b):
        be.set_tasks(*tasks)
        be.reset()
    g = a.b.get_plane("grid")
    assert g.shape == (N, H, W) and torch.equal(g, a.b.plane("grid"))
    host = torch.empty((N, H, W), dtype=torch.int8).pin_memory()
    a.b.get_plane("input", host)
    torch.cuda.synchronize()
    assert np.array_equal(host.numpy(), tasks[0])
    new = torch.from_numpy(rng.integers(0, 10, (N, H, W)).astype(np.int8))
    for be in (a, b):
        be.b.set_plane("grid", new.cuda())
    assert torch.equal(a.b.plane("grid"), new.cuda()) and a.padding_is_zero()
    for form in ("mask", "bits", "point"):
        if form == "point":
            pay = torch.from_numpy(np.stack([rng.integers(0, H, (K, N)), rng.integers(0, W, (K, N))], -1).astype(np.int32)).cuda()
        else:
            pay = torch.from_numpy((rng.random((K, N, H, W)) < 0.1).astype(np.int8)).cuda()
        op = torch.from_numpy(rng.integers(0, 35, (K, N)).astype(np.int32)).cuda()
        rs, ts = [], []
        for i in range(K):
            if form == "mask":
                r, t = a.b.step_mask(pay[i], op[i], 1)
            elif form == "bits":
                r, t = a.b.step_bits(a.b.pack_mask_bits(pay[i]), op[i], 1)
            else:
                r, t = a.b.step_point(pay[i], op[i], 1)
            rs.append(r.clone()), ts.append(t.clone())
        many = torch.stack([b.b.pack_mask_bits(pay[i]) for i in range(K)]) if form == "bits" else pay
        r2, t2 = b.b.step_many(form, many.contiguous(), op, 1)
        assert torch.equal(torch.stack(rs), r2) and torch.equal(torch.stack(ts), t2), form
        for k in a.b.planes:
            assert torch.equal(a.b.planes[k], b.b.planes[k]), (form, k)
        assert torch.equal(a.b.rec, b.b.rec) and torch.equal(a.b.cnt, b.b.cnt)
    assert a.status() == b.status()


def test_vec_transition_in_place_walks_trajectories():
    """ARCVecEnv.transition(in_place=True): the rows buffer is advanced by one action per call, only changed planes rewritten — after
    T calls it equals T out-of-place transitions and T resident steps of the same envs."""
    import torch
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    N, T = 256, 12
    v = ARCVecEnv(O2ARCv2Env, N, _loader(), max_grid_size=(12, 12), max_trial=3, seed=2)
    v.reset()
    rows0 = v.state_rows().clone()
    bb, op = _actions(T, N, 12, 12, 8)
    walk, _, _ = v.transition(rows0, {"bbox": bb[0], "operation": op[0]})   # the first hop allocates the [N, stride] buffer
    branch = walk.clone()
    for i in range(1, T):
        walk2, r_ip, t_ip = v.transition(walk, {"bbox": bb[i], "operation": op[i]}, in_place=True)
        assert walk2.data_ptr() == walk.data_ptr()
        branch, r_op, t_op = v.transition(branch.contiguous(), {"bbox": bb[i], "operation": op[i]})
        assert torch.equal(r_ip, r_op) and torch.equal(t_ip, t_op), i
        assert torch.equal(walk2[:, :branch.shape[1]], branch), i
    for i in range(T):
        v.step_bbox(bb[i], op[i])
    assert torch.equal(v.state_rows(), walk[:, :rows0.shape[1]])
    v.check_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8192, 1024, 960])
def test_step_many_ordered_dispatch_is_scheduling_only(n):
    """arcle_step_many with the dispatch order on (since round 5 every launch orders itself inside groups of 32 envs) and off
    (arcle_set_dispatch_order): rewards, terminated flags and every byte of state are equal — bbox + op arrays and 5-tuple records,
    eager and replayed as a hipGraph with the action buffers rewritten in between, bad op indices included.  (n = 960 is not a multiple
    of 256 and n = 1024 is below the window: the library steps those in plain env order, silently.)"""
    import torch
    import bench
    K = 30
    dev = torch.device("cuda:0")
    bb_np, op_np = bench.make_actions(K, n, 123)
    op_np = op_np.copy()
    op_np[3, ::97] = 35 + (np.arange(len(op_np[3, ::97])) % 40)  # indices past the table (raise ARCLE_ST_BAD_OP, step skipped)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    act5 = torch.cat([bb, op[:, :, None]], 2).contiguous()
    for form, pay, o in (("bbox", bb, op), ("bbox5", act5, None)):
        a, b = bench.make_batch(dev, n, seed=9), bench.make_batch(dev, n, seed=9)
        a.set_dispatch_order(False)
        b.set_dispatch_order(True)
        FL = a.elide_flag | 1
        ra, ta = a.step_many(form, pay, o, FL)
        rb, tb = b.step_many(form, pay, o, FL)
        torch.cuda.synchronize()
        assert torch.equal(ra, rb) and torch.equal(ta, tb), form
        for k in a.planes:
            assert torch.equal(a.planes[k], b.planes[k]), (form, k)
        assert torch.equal(a.rec, b.rec) and torch.equal(a.cnt, b.cnt) and a.status() == b.status() == 1, form  # (1 = ARCLE_ST_BAD_OP)
        # replayed as a graph, with new actions written into the captured buffers between replays
        g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=st):
            b.step_many(form, pay, o, FL, rb, tb)
        for seed in (124, 125):
            bb2, op2 = bench.make_actions(K, n, seed)
            bb.copy_(torch.from_numpy(bb2)), op.copy_(torch.from_numpy(op2))
            if form == "bbox5":
                act5.copy_(torch.cat([bb, op[:, :, None]], 2))
            ra, ta = a.step_many(form, pay, o, FL)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(ra, rb) and torch.equal(ta, tb), (form, seed)
            assert torch.equal(a.planes["grid"], b.planes["grid"]) and torch.equal(a.rec, b.rec), (form, seed)
        bb.copy_(torch.from_numpy(bb_np)), op.copy_(torch.from_numpy(op_np))
