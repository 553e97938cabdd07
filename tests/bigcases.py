"""Test bodies shared by tests/test_big_emu.py (the workgroup-per-env kernel bodies on host threads) and tests/test_big_hip.py (the same
kernels on the GPU through the C ABI): grids of more than 1024 cells against the oracle."""
import numpy as np

import backends as B
from oracle import oracle as O


def _stepped(be_cls, kind, H, W, N, S, seed, flags=0):
    rng = np.random.default_rng(seed)
    ops = O.KIND_OPS[kind]()
    be, orc = be_cls(N, H, W, 3, kind, ops), B.OracleBackend(N, H, W, 3, kind, ops)
    inp = np.zeros((N, H, W), np.int8)
    dims = np.zeros((N, 2), np.int8)
    for n in range(N):
        h, w = rng.integers(2, H + 1), rng.integers(2, W + 1)
        inp[n, :h, :w] = rng.integers(0, 10, (h, w))
        dims[n] = (h, w)
    for b_ in (be, orc):
        b_.set_tasks(inp, dims, inp, dims)
        b_.reset()
    return be, orc, rng, ops


def rows_case(be_cls, kind):
    """Flat rows (full / FilterO2ARC), fused rows with the step-output tail, packed gather rows and the state-row round trip — against rows
    built on the host from the oracle's fields in the layout the one-wavefront kernels are pinned on (backends.row_layout)."""
    H, W, N = 36, 41, 4
    be, orc, rng, ops = _stepped(be_cls, kind, H, W, N, 0, 5)
    be.set_flat_output(filtered=False, tail=True)
    be.set_packed_output()
    for s in range(12):
        op = rng.integers(0, len(ops), N).astype(np.int32)
        bbox = np.stack([rng.integers(0, H, N), rng.integers(0, W, N), rng.integers(0, H, N), rng.integers(0, W, N)], 1)
        r1, t1 = be.step("bbox", bbox, op, 128 | 256)
        r2, t2 = orc.step("bbox", bbox, op, 0)
        want = B.state_rows(orc)
        assert np.array_equal(be.fused_flat(), want), f"fused rows differ at step {s}"
        tail = be.fused_tail()
        assert np.array_equal(tail[:, 0], r2) and np.array_equal(tail[:, 1], orc.counters()[:, 0]) and np.array_equal(tail[:, 2], orc.counters()[:, 1])
        assert np.array_equal(tail[:, 3] & 0xFF, t2)
        pk = be.fused_packed()
        P = H * W
        assert np.array_equal(pk[:, :P].view(np.int8), orc.get("grid").reshape(N, P))
        assert np.array_equal(pk[:, P:P + 2].view(np.int8), orc.get("grid_dim"))
        assert np.array_equal(pk[:, P + 2:P + 6].copy().view(np.int32)[:, 0], r2) and np.array_equal(pk[:, P + 6], t2)
        assert not pk[:, P + 7:].any()
    assert np.array_equal(be.flat_obs(False), B.state_rows(orc))
    assert np.array_equal(be.packed_obs(), be.fused_packed())
    if kind == "o2arc":
        lay = dict(B.row_layout(kind, H * W))
        full = B.state_rows(orc)
        off, segs = 0, {}
        for f, n in B.row_layout(kind, H * W):
            segs[f] = full[:, off:off + n]
            off += n
        want = np.concatenate([segs[f] for f in ("active", "clip", "clip_dim", "grid", "grid_dim", "object", "object_dim", "object_pos", "trials_remain")], 1)
        assert np.array_equal(be.flat_obs(True), want) and lay["grid"] == H * W
    # state rows in: a fresh backend takes the rows and holds the same state
    be2 = be_cls(N, H, W, 3, kind, ops)
    be2.set_state_rows(B.state_rows(orc))
    for f, _ in B.row_layout(kind, H * W):
        assert np.array_equal(be2.get(f), orc.get(f)), f
    assert be2.padding_is_zero()


def truncation_case(be_cls):
    H, W, N = 40, 40, 4
    be, orc, rng, ops = _stepped(be_cls, "o2arc", H, W, N, 0, 6)
    be.set_truncation(5)
    for s in range(14):
        op = rng.integers(0, len(ops), N).astype(np.int32)
        xy = np.stack([rng.integers(0, H, N), rng.integers(0, W, N)], 1)
        be.step("point", xy, op, 1 | 4)
        steps = be.counters()[:, 0]
        assert np.array_equal(np.asarray(be.trunc), (steps >= 5).astype(np.uint8))
        assert steps.max() <= 5  # an env that ran out of steps restarts on its next step


def task_table_case(be_cls):
    """reset_from_table / reset_sampled: the table entry lands in the planes, and the device draw is the function of (seed, global env id,
    episode) the host mirrors (arcle_amd.sampling) — the same one the one-wavefront kernels use."""
    from arcle_amd import sampling
    H, W, N, T = 40, 40, 6, 5
    rng = np.random.default_rng(3)
    ins = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
    outs = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
    be = be_cls(N, H, W, 3, "o2arc", O.o2arc_ops())
    be.set_task_table(ins, outs)
    idx = np.array([0, 4, 2, 2, 9, 1], np.int32)  # entry 9 does not exist: BAD_TASK, env untouched
    be.reset_from_table(idx)
    assert be.status() == 4
    for n, t in enumerate(idx):
        if t >= T:
            assert not be.get("grid")[n].any()
            continue
        a, b_ = ins[t], outs[t]
        assert np.array_equal(be.get("grid")[n, :a.shape[0], :a.shape[1]], a) and np.array_equal(be.get("input")[n], be.get("grid")[n])
        assert tuple(be.get("input_dim")[n]) == a.shape and tuple(be.get("answer_dim")[n]) == b_.shape
        assert np.array_equal(be.get("answer")[n, :b_.shape[0], :b_.shape[1]], b_)
    assert be.padding_is_zero()
    off, cnt = np.array([0, 2, 3], np.int32), np.array([2, 1, 2], np.int32)
    be.set_sampler(off, cnt, seed=1234, env_base=100)
    be.reset_sampled()
    be.reset_sampled(np.array([1, 0, 1, 0, 0, 0], np.uint8))
    episode, cur_task = np.asarray(be.episode), np.asarray(be.cur_task)
    assert episode.tolist() == [2, 1, 2, 1, 1, 1]
    for n in range(N):
        ep = int(episode[n]) - 1
        prob, sub, _, _ = sampling.draw_task(1234, 100 + n, ep, cnt)
        assert int(cur_task[n]) == off[prob] + sub


def _augmented(a, k, perm):
    """np.rot90 of the colour-permuted un-padded grid (agents/env.py:31-42)."""
    lut = np.arange(256, dtype=np.int64)
    lut[:10] = perm
    return np.rot90(lut[a.astype(np.uint8)].astype(np.int8), k)


def aug_case(be_cls):
    """Task augmentation on the big path: explicit (rot90 count + colour permutation per env) and device-drawn (the function of (seed, global
    env id, episode) arcle_amd.sampling mirrors), on a square plane; on a non-square one an explicit quarter turn that does not fit is
    refused (AUG_DOMAIN, env untouched) and a drawn one is dropped (k & 2)."""
    from arcle_amd import sampling
    for (H, W) in ((40, 40), (34, 45)):
        N, T = 6, 5
        rng = np.random.default_rng(H + W)
        ins = [rng.integers(0, 10, (rng.integers(2, H + 1), rng.integers(2, W + 1))).astype(np.int8) for _ in range(T)]
        outs = [rng.integers(0, 10, (rng.integers(2, H + 1), rng.integers(2, W + 1))).astype(np.int8) for _ in range(T)]
        if H != W:  # one entry that only fits unturned, one that fits both ways
            ins[0], outs[0] = ins[0][:, :W][:H], np.zeros((3, W), np.int8) + 4
            ins[1], outs[1] = ins[1][:H, :H][:20, :25], outs[1][:20, :30]
        fits = [a.shape[1] <= H and a.shape[0] <= W and b_.shape[1] <= H and b_.shape[0] <= W for a, b_ in zip(ins, outs)]
        be = be_cls(N, H, W, 3, "o2arc", O.o2arc_ops())
        be.set_task_table(ins, outs)
        idx = rng.integers(0, T, N).astype(np.int32)
        idx[0], idx[1] = 0, 1
        k = rng.integers(0, 4, N).astype(np.uint8)
        k[0] = 1
        perm = np.stack([rng.permutation(10) for _ in range(N)]).astype(np.uint8)
        be.reset()  # (zeros: an env the explicit reset must leave untouched stays recognisable)
        be.reset_from_table(idx, None, k, perm)
        st = be.status()
        refused = [n for n in range(N) if (k[n] & 1) and not fits[idx[n]]]
        assert (st == 16) == bool(refused), (st, refused)
        for n in range(N):
            if n in refused:
                assert not be.get("grid")[n].any() and not be.get("input")[n].any()
                continue
            a, b_ = _augmented(ins[idx[n]], int(k[n]), perm[n]), _augmented(outs[idx[n]], int(k[n]), perm[n])
            want = np.zeros((H, W), np.int8)
            want[:a.shape[0], :a.shape[1]] = a
            assert np.array_equal(be.get("input")[n], want) and np.array_equal(be.get("grid")[n], want), (H, W, n)
            want = np.zeros((H, W), np.int8)
            want[:b_.shape[0], :b_.shape[1]] = b_
            assert np.array_equal(be.get("answer")[n], want)
            assert tuple(be.get("input_dim")[n]) == a.shape and tuple(be.get("answer_dim")[n]) == b_.shape and tuple(be.get("grid_dim")[n]) == a.shape
        assert be.padding_is_zero()
        off, cnt = np.array([0, 2, 3], np.int32), np.array([2, 1, 2], np.int32)
        be.set_sampler(off, cnt, seed=99, env_base=7, aug_flags=3)
        be.reset_sampled()
        assert be.status() == 0
        cur = np.asarray(be.cur_task)
        for n in range(N):
            prob, sub, dk, dperm = sampling.draw_task(99, 7 + n, 0, cnt, 3)
            t = int(off[prob] + sub)
            assert int(cur[n]) == t
            if (dk & 1) and not fits[t]:
                dk &= 2
            a = _augmented(ins[t], dk, dperm)
            want = np.zeros((H, W), np.int8)
            want[:a.shape[0], :a.shape[1]] = a
            assert np.array_equal(be.get("grid")[n], want), (H, W, n, dk)
            assert tuple(be.get("answer_dim")[n]) == _augmented(outs[t], dk, dperm).shape
