"""Backend-independent checks of the round-3 boundary additions: the BBoxWrapper 5-tuple record and bit-packed mask ingress
forms, state rows in (arcle_set_state_rows), the stateless batched transition (arcle_transition_rows), the step-output tail of a
flat row, dense reward on auto-reset steps, and the kernel's byte accounting.  As in tests/features.py every function takes a
backend class (tests/backends.py: EmuBackend on the CPU, HipBackend on the GPU) and returns a list of mismatch strings; the
reference point is always the oracle (or rows already pinned on the reference's FlattenObservation by features.flat)."""
import numpy as np

import backends as B
from oracle import oracle as O

STEP_AUTORESET, STEP_ELIDE, STEP_TRUNCATE, STEP_DENSE, STEP_ROS, STEP_FLAT_OBS, STEP_ROWS_INC = 1, 2, 4, 16, 64, 128, 512


def _tasks(rng, N, H, W, same_answer=0.5):
    inp, ans = np.zeros((N, H, W), np.int8), np.zeros((N, H, W), np.int8)
    idim, adim = np.zeros((N, 2), np.int8), np.zeros((N, 2), np.int8)
    for n in range(N):
        ih, iw = rng.integers(1, H + 1), rng.integers(1, W + 1)
        g = rng.integers(0, 4, (ih, iw)).astype(np.int8)
        inp[n, :ih, :iw], idim[n] = g, (ih, iw)
        if rng.random() < same_answer:
            ans[n, :ih, :iw], adim[n] = g, (ih, iw)
        else:
            ah, aw = rng.integers(1, H + 1), rng.integers(1, W + 1)
            ans[n, :ah, :aw], adim[n] = rng.integers(0, 10, (ah, aw)), (ah, aw)
    return inp, idim, ans, adim


def _random_actions(rng, N, H, W, n_ops):
    """One action per env in all three classic forms' worth of variety -> (ingress, payload, op)."""
    ing = ["bbox", "point", "mask"][rng.integers(0, 3)]
    op = rng.integers(0, n_ops, N).astype(np.int32)
    op[rng.random(N) < 0.1] = n_ops - 1
    if ing == "bbox":
        pay = np.stack([rng.integers(0, H, N), rng.integers(0, W, N), rng.integers(0, H, N), rng.integers(0, W, N)], 1).astype(np.int32)
        small = rng.random(N) < 0.5
        pay[small, 2] = np.minimum(H - 1, pay[small, 0] + rng.integers(0, 4, small.sum()))
        pay[small, 3] = np.minimum(W - 1, pay[small, 1] + rng.integers(0, 4, small.sum()))
    elif ing == "point":
        pay = np.stack([rng.integers(0, H, N), rng.integers(0, W, N)], 1).astype(np.int32)
    else:
        pay = np.zeros((N, H, W), np.int8)
        for n in range(N):
            t = rng.integers(0, 4)
            if t == 1:
                pay[n] = rng.random((H, W)) < rng.random() * 0.3
            elif t == 2:
                pay[n, rng.integers(0, H), rng.integers(0, W)] = 1
            elif t == 3:
                x, y = rng.integers(0, H), rng.integers(0, W)
                pay[n, x:x + rng.integers(1, 5), y:y + rng.integers(1, 5)] = 1
    return ing, pay, op


def _pair(cls, N, H, W, seed, max_trial=3, kind="o2arc", ops=None, warm=10):
    """A backend and the oracle in the same (non-trivial) state: same tasks, `warm` identical random steps."""
    rng = np.random.default_rng(seed)
    ops = ops or O.KIND_OPS[kind]()
    be, orc = cls(N, H, W, max_trial, kind, ops), B.OracleBackend(N, H, W, max_trial, kind, ops)
    tasks = _tasks(rng, N, H, W)
    for b in (be, orc):
        b.set_tasks(*tasks)
        b.reset()
    for _ in range(warm):
        ing, pay, op = _random_actions(rng, N, H, W, len(ops))
        op[op == len(ops) - 1] = 0  # (no submits while warming up: keep the envs alive)
        be.step(ing, pay, op)
        orc.step(ing, pay, op)
    be.status(), orc.status()
    return be, orc, rng, ops


def _state_fields(kind):
    return [f for f in O.PLANES[:-1] if f in O.KIND_PLANES[kind]] + [
        f for f in O.REC if f != "answer_dim" and (kind == "o2arc" or f in ("input_dim", "grid_dim", "trials_remain", "terminated")
                                                   or (kind == "arc" and f == "clip_dim"))]


def new_ingress_forms(cls):
    """bbox5 records and bit-packed masks step exactly like bbox + op / int8 masks (random differential traces vs the oracle)."""
    errs = []
    for H, W, seed in ((30, 30, 1), (12, 12, 2), (7, 12, 3), (5, 40, 4)):
        errs += B.random_trace_compare(cls, "o2arc", O.o2arc_ops(), H, W, N=6, S=40, seed=seed, max_trial=3, new_forms=True,
                                       op_weights=[1] * 10 + [2] * 10 + [3] * 8 + [2] * 7)
    errs += B.random_trace_compare(cls, "o2arc", O.o2arc_ops(), 30, 30, N=6, S=30, seed=9, flags=STEP_AUTORESET | STEP_ELIDE, new_forms=True)
    errs += B.random_trace_compare(cls, "arc", O.arc_ops(), 30, 30, N=4, S=30, seed=5, new_forms=True)
    return errs


def mask_bits_packer(cls):
    """arcle_pack_mask_bits == np.packbits of (mask != 0), any int8 values, odd grid sizes."""
    errs = []
    for H, W in ((30, 30), (7, 12), (32, 32), (5, 5)):
        rng = np.random.default_rng(H * 100 + W)
        N = 9
        be = cls(N, H, W, -1, "o2arc", O.o2arc_ops())
        m = rng.integers(-2, 3, (N, H, W)).astype(np.int8) * (rng.random((N, H, W)) < 0.4)
        m[0] = 0
        m[1] = 1
        got = be.pack_mask_bits(m)
        if not np.array_equal(got, B.pack_bits(m)):
            errs.append(f"{H}x{W}: packed bit rows differ")
    return errs


def state_rows_roundtrip(cls):
    """get_state_rows (the pinned flat writer) -> set_state_rows on a fresh batch reproduces every state field; masked ingest leaves
    the other envs alone; rows built on the host from the ORACLE's state ingest to the oracle's state."""
    errs = []
    for kind, H, W in (("o2arc", 30, 30), ("o2arc", 7, 12), ("arc", 30, 30), ("raw", 5, 5)):
        N = 7
        be, orc, rng, ops = _pair(cls, N, H, W, seed=H + W, kind=kind, warm=14)
        rows = be.flat_obs(False)
        if not np.array_equal(rows, B.state_rows(orc)):
            errs.append(f"{kind} {H}x{W}: flat rows differ from the rows built from the oracle's state")
        fresh = cls(N, H, W, 3, kind, ops)
        fresh.set_tasks(orc.get("input"), orc.get("input_dim"), orc.get("answer"), orc.get("answer_dim"))
        fresh.reset()
        padded = np.zeros((N, rows.shape[1] + 5), np.int8)  # an odd stride: rows at arbitrary alignment
        padded[:, :rows.shape[1]] = rows
        mask = (np.arange(N) % 3 != 1).astype(np.uint8)
        fresh.set_state_rows(padded, mask)
        for f in _state_fields(kind):
            a, b = fresh.get(f), orc.get(f)
            keep = mask.astype(bool)
            if not np.array_equal(a[keep], b[keep]):
                errs.append(f"{kind} {H}x{W}: field {f} differs after set_state_rows")
        if not np.array_equal(fresh.get("grid")[~mask.astype(bool)], fresh.get("input")[~mask.astype(bool)]):
            errs.append(f"{kind} {H}x{W}: a masked-out env was touched by set_state_rows")
        if hasattr(fresh, "padding_is_zero") and not fresh.padding_is_zero():
            errs.append(f"{kind} {H}x{W}: plane padding not zero after set_state_rows")
        # the restored envs continue exactly like the oracle
        fresh.set_state_rows(padded)
        for _ in range(6):
            ing, pay, op = _random_actions(rng, N, H, W, len(ops))
            r1, t1 = fresh.step(ing, pay, op)
            r2, t2 = orc.step(ing, pay, op)
            if not (np.array_equal(r1, r2) and np.array_equal(t1, t2)):
                errs.append(f"{kind} {H}x{W}: reward / terminated differ after the restore")
        fresh.status(), orc.status()
        for f in _state_fields(kind):
            if not np.array_equal(fresh.get(f), orc.get(f)):
                errs.append(f"{kind} {H}x{W}: field {f} diverged after the restore")
    return errs


def transition_rows(cls, cases=(("o2arc", 30, 30, 3), ("o2arc", 7, 12, -1), ("o2arc", 12, 12, 1), ("arc", 30, 30, 3), ("raw", 5, 5, 2))):
    """arcle_transition_rows(rows, actions) == oracle.step on the same states, for every ingress form; the resident envs stay
    untouched; src_env picks the answer; the tail carries (reward, 1, submit counted, terminated, status)."""
    errs = []
    for kind, H, W, mt in cases:
        N = 8
        be, orc, rng, ops = _pair(cls, N, H, W, seed=H * W + mt, max_trial=mt, kind=kind, warm=12)
        L = sum(n for _, n in B.row_layout(kind, H * W))
        for rep in range(6):
            rows = B.state_rows(orc)
            before = {f: be.get(f) for f in _state_fields(kind)}
            cnt_before = be.counters()
            ing, pay, op = _random_actions(rng, N, H, W, len(ops))
            if rep == 3:
                op[0] = len(ops) + 2  # an out-of-range op: the row passes through, status bit in the tail
            trials_before = orc.get("trials_remain")[:, 0].copy()
            out, r1, t1 = be.transition_rows(rows, ing, pay, op, tail=True, in_place=(rep % 2 == 1))  # (odd rounds: rows_out IS rows_in)
            r2, t2 = orc.step(ing, pay, op)
            want = B.state_rows(orc)
            tag = f"{kind} {H}x{W} rep {rep} {ing}"
            if not np.array_equal(out[:, :L], want):
                bad = np.nonzero((out[:, :L] != want).any(1))[0]
                errs.append(f"{tag}: output rows differ for rows {bad.tolist()} (ops {op[bad].tolist()})")
            if out[:, L:(L + 15) & ~15].any():
                errs.append(f"{tag}: row padding not zero")
            if not (np.array_equal(r1, r2) and np.array_equal(t1, t2)):
                errs.append(f"{tag}: reward / terminated differ")
            tail = np.ascontiguousarray(out[:, -16:]).view(np.int32)
            submit = (op == len(ops) - 1) & (trials_before != 0)
            if not (np.array_equal(tail[:, 0], r2) and np.array_equal(tail[:, 3] & 0xff, t2)):
                errs.append(f"{tag}: tail reward / terminated differ")
            ost = orc.status()
            st_env = (tail[:, 3] >> 16) & 0xff
            if int(np.bitwise_or.reduce(st_env)) != ost or be.status() != ost:
                errs.append(f"{tag}: per-row status {st_env.tolist()} vs oracle status {ost}")
            ok = st_env == 0
            if not np.array_equal(tail[ok, 2], submit[ok].astype(np.int32)) or not np.array_equal(tail[ok, 1], np.ones(ok.sum(), np.int32)):
                errs.append(f"{tag}: tail counters {tail[:, 1:3].tolist()} (submits expected {submit.tolist()})")
            for f, v in before.items():
                if not np.array_equal(be.get(f), v):
                    errs.append(f"{tag}: resident field {f} was touched by transition_rows")
            if not np.array_equal(be.counters(), cnt_before):
                errs.append(f"{tag}: resident counters were touched")
            if len(errs) > 10:
                return errs
        # more rows than envs: every row names the env whose answer it is judged against
        M = 2 * N + 3
        src = rng.integers(0, N, M).astype(np.int32)
        base_rows = B.state_rows(orc)
        rows = base_rows[src]
        ing, pay, op = _random_actions(rng, M, H, W, len(ops))
        op[: M // 2] = len(ops) - 1  # submits: the answer matters
        out, r1, t1 = be.transition_rows(rows, ing, pay, op, src_env=src)
        big = B.OracleBackend(M, H, W, mt, kind, ops)
        big.set_tasks(orc.get("input")[src], orc.get("input_dim")[src], orc.get("answer")[src], orc.get("answer_dim")[src])
        big.reset()
        for f in _state_fields(kind):
            (big.env.planes[f] if f in big.env.planes else big.env.field(f))[:] = orc.get(f)[src]
        r2, t2 = big.step(ing, pay, op)
        if not (np.array_equal(out[:, :L], B.state_rows(big)) and np.array_equal(r1, r2) and np.array_equal(t1, t2)):
            errs.append(f"{kind} {H}x{W}: src_env transition differs from the oracle")
        be.status(), big.status()
    return errs


def flat_tail(cls):
    """STEP_FLAT_OBS rows with the 16-byte tail: (reward, action_steps, submit_count, terminated, truncated, per-env status)."""
    errs = []
    N, H, W = 8, 12, 12
    be, orc, rng, ops = _pair(cls, N, H, W, seed=77, max_trial=2, warm=5)
    be.set_flat_output(False, tail=True)
    be.set_truncation(9)
    for s in range(14):
        ing, pay, op = _random_actions(rng, N, H, W, len(ops))
        if s == 6:
            op[2] = 40
        r1, t1 = be.step(ing, pay, op, STEP_FLAT_OBS | STEP_TRUNCATE)
        r2, t2 = orc.step(ing, pay, op)
        tail, cnt = be.fused_tail(), orc.counters()
        ost = orc.status()
        be.status()
        want = [r2, cnt[:, 0], cnt[:, 1], t2, (cnt[:, 0] >= 9).astype(np.int32)]
        got = [tail[:, 0], tail[:, 1], tail[:, 2], tail[:, 3] & 0xff, (tail[:, 3] >> 8) & 0xff]
        for name, a, b in zip(("reward", "steps", "submit_count", "terminated", "truncated"), got, want):
            if not np.array_equal(a, b):
                errs.append(f"step {s}: tail {name} {a.tolist()} != {b.tolist()}")
        if int(np.bitwise_or.reduce((tail[:, 3] >> 16) & 0xff)) != ost:
            errs.append(f"step {s}: tail status bits vs oracle status {ost}")
        if not np.array_equal(be.fused_flat(), B.state_rows(orc)):
            errs.append(f"step {s}: fused rows differ from the oracle's state rows")
    return errs


def dense_on_autoreset(cls, H=10, W=10):
    """ARCLE_STEP_DENSE with auto-reset: the step that re-initialises an env (and a skipped step) reports the pair (0, 0) = no dense
    term, never the previous step's pair; every executed step reports the pair of the state it produced."""
    errs = []
    N = 10
    rng = np.random.default_rng(5)
    ops = O.o2arc_ops()
    be, orc = cls(N, H, W, 1, "o2arc", ops), B.OracleBackend(N, H, W, 1, "o2arc", ops)
    tasks = _tasks(rng, N, H, W, same_answer=0.7)
    for b in (be, orc):
        b.set_tasks(*tasks)
        b.reset()
    be.set_dense_output()
    ended = np.zeros(N, bool)
    for s in range(30):
        ing, pay, op = _random_actions(rng, N, H, W, len(ops))
        op[rng.random(N) < 0.3] = 34
        if s == 11:
            op[1] = 99
        r1, t1 = be.step(ing, pay, op, STEP_AUTORESET | STEP_DENSE)
        r2, t2 = orc.step(ing, pay, op, STEP_AUTORESET)
        if not (np.array_equal(r1, r2) and np.array_equal(t1, t2)):
            errs.append(f"step {s}: reward / terminated differ")
        d = be.dense
        skipped = ended | ((op >= 35) & ~ended)
        gh, ah = orc.get("grid_dim").astype(int), orc.get("answer_dim").astype(int)
        g, a = orc.get("grid"), orc.get("answer")
        for n in range(N):
            if skipped[n]:
                if tuple(d[n]) != (0, 0):
                    errs.append(f"step {s} env {n}: reset / skipped step reports dense {tuple(d[n])}, expected (0, 0)")
                continue
            mh, mw = min(gh[n, 0], ah[n, 0]), min(gh[n, 1], ah[n, 1])
            correct = int((g[n, :mh, :mw] == a[n, :mh, :mw]).sum())
            G, A = gh[n, 0] * gh[n, 1], ah[n, 0] * ah[n, 1]
            total = mh * mw + (abs(A - G) if (gh[n, 0] <= ah[n, 0]) == (gh[n, 1] <= ah[n, 1]) else
                               abs(gh[n, 0] - ah[n, 0]) * mw + abs(gh[n, 1] - ah[n, 1]) * mh)
            if tuple(d[n]) != (correct, total):
                errs.append(f"step {s} env {n}: dense {tuple(d[n])} != {(correct, total)}")
        ended = t2.astype(bool)
        be.status(), orc.status()
        if len(errs) > 10:
            break
    return errs


def incremental_rows(cls):
    """ARCLE_STEP_ROWS_INCREMENTAL: rows kept across steps (only the segments of stored planes rewritten) stay byte-identical to the
    stand-alone writer's rows of the same state — full and FilterO2ARC layouts, with auto-reset and the zero-fill elision on."""
    errs = []
    for filtered in (False, True):
        for H, W, flags in ((30, 30, STEP_AUTORESET | STEP_ELIDE), (12, 12, 0), (7, 12, STEP_AUTORESET)):
            N = 8
            rng = np.random.default_rng(H + W + filtered)
            ops = O.o2arc_ops()
            be = cls(N, H, W, 2, "o2arc", ops)
            be.set_tasks(*_tasks(rng, N, H, W, same_answer=0.8))
            be.reset()
            be.set_flat_output(filtered)
            be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 32, np.int32), flags | STEP_FLAT_OBS)  # one full write
            for s in range(40):
                ing, pay, op = _random_actions(rng, N, H, W, len(ops))
                op[rng.random(N) < 0.2] = 34
                be.step(ing, pay, op, flags | STEP_FLAT_OBS | STEP_ROWS_INC)
                if not np.array_equal(be.fused_flat(), be.flat_obs(filtered)):
                    bad = np.nonzero((be.fused_flat() != be.flat_obs(filtered)).any(1))[0]
                    errs.append(f"{H}x{W} filtered={filtered} step {s}: incremental rows differ for envs {bad.tolist()} (ops {op[bad].tolist()})")
                    break
            be.status()
    return errs


def dense_cache(cls):
    """The dense pair is recomputed only when a step stored the grid; the cached pairs equal a recomputation from the state after
    EVERY step — incl. steps that leave the grid alone, resets through every path, and state ingest."""
    errs = []
    N, H, W = 12, 10, 10
    rng = np.random.default_rng(11)
    ops = O.o2arc_ops()
    be, orc = cls(N, H, W, 2, "o2arc", ops), B.OracleBackend(N, H, W, 2, "o2arc", ops)
    tasks = _tasks(rng, N, H, W, same_answer=0.6)
    for b in (be, orc):
        b.set_tasks(*tasks)
        b.reset()
    be.set_dense_output()

    def expect(n):
        gh, ah = orc.get("grid_dim").astype(int)[n], orc.get("answer_dim").astype(int)[n]
        g, a = orc.get("grid")[n], orc.get("answer")[n]
        mh, mw = min(gh[0], ah[0]), min(gh[1], ah[1])
        correct = int((g[:mh, :mw] == a[:mh, :mw]).sum())
        G, A = gh[0] * gh[1], ah[0] * ah[1]
        total = mh * mw + (abs(A - G) if (gh[0] <= ah[0]) == (gh[1] <= ah[1]) else abs(gh[0] - ah[0]) * mw + abs(gh[1] - ah[1]) * mh)
        return (correct, total)
    for s in range(60):
        ing, pay, op = _random_actions(rng, N, H, W, len(ops))
        quiet = rng.random(N) < 0.5
        op[quiet] = rng.choice([10, 28, 29, 20], quiet.sum())  # mostly grid-preserving: failed fills, copies, no-op moves
        if s % 13 == 5:  # a reset from outside, masked
            m = (rng.random(N) < 0.4).astype(np.uint8)
            be.reset(m)
            orc.reset(m)
        if s % 17 == 9:  # state ingest: all envs back to an earlier state
            rows = B.state_rows(orc)
            orc.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 3, np.int32))
            be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 3, np.int32), STEP_DENSE)
            be.set_state_rows(rows)
            lay, off = B.row_layout("o2arc", H * W), 0
            for f, n in lay:
                dst = orc.env.planes[f] if f in orc.env.planes else orc.env.field(f)
                dst[:] = rows[:, off:off + n].reshape(dst.shape)
                off += n
        if s % 7 == 3:  # a step WITHOUT the dense flag moves grids behind the cache's back: the launcher drops the cache for it
            ing0, pay0, op0 = _random_actions(rng, N, H, W, len(ops))
            op0[:] = rng.integers(0, 10, N)  # Color: every grid changes
            be.step(ing0, pay0, op0, 0)
            orc.step(ing0, pay0, op0)
        r1, t1 = be.step(ing, pay, op, STEP_DENSE)
        r2, t2 = orc.step(ing, pay, op)
        if not (np.array_equal(r1, r2) and np.array_equal(t1, t2)):
            errs.append(f"step {s}: reward / terminated differ")
        d = be.dense
        for n in range(N):
            if tuple(d[n]) != expect(n):
                errs.append(f"step {s} env {n} op {op[n]}: dense {tuple(d[n])} != {expect(n)}")
        be.status(), orc.status()
        if len(errs) > 8:
            break
    return errs


def partial_store_invariants(cls):
    """The invariant ARCLE_STEP_ELIDE_SELECTED rests on (an inactive env holds an all-zero `selected` plane) under stress: tables with
    Rotate 180 and the Flip D0 / D1 quirk (tile transposed, object_dim not), object ops dominating, objects walked off the grid (int8
    wrap of object_pos), every ingress form, auto-reset — every plane compared with the oracle after every step.  (Written for an
    experiment that also limited the stores of `selected` / `object` / `object_sel` to the rows that can differ — 22 % fewer bytes
    issued, but slower: profiles/round3_experiments.txt — and kept: it catches a wrong row bound within a few steps.)"""
    errs = []
    ops = O.o2arc_ops()
    ops[24] = O.desc(O.OP_ROTATE, 2)
    ops[26] = O.desc(O.OP_FLIP, 2)   # D0
    ops[27] = O.desc(O.OP_FLIP, 3)   # D1
    heavy = [1] * 20 + [6] * 8 + [2] * 7
    for (H, W), seed in (((30, 30), 1), ((16, 16), 2), ((20, 30), 3), ((32, 32), 4)):
        for flags in (STEP_ELIDE, STEP_ELIDE | STEP_AUTORESET):
            errs += B.random_trace_compare(cls, "o2arc", ops, H, W, N=6, S=70, seed=seed + 10 * flags, max_trial=2, flags=flags, op_weights=heavy)
    errs += B.random_trace_compare(cls, "o2arc", O.o2arc_ops(), 30, 30, N=6, S=120, seed=99, flags=STEP_ELIDE, op_weights=[1] * 20 + [12] * 4 + [2] * 4 + [1] * 7)
    return errs
