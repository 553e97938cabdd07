"""Backend-independent checks of the callers / data formats either side of the hot path against the
golden vectors captured from the reference by tests/golden/make_golden_research.py.  Each function takes a backend class
(tests/backends.py: EmuBackend for the CPU suite, HipBackend for the GPU suite) and returns a list of mismatch strings."""
import json
import os

import numpy as np

import backends as B
from oracle import oracle as O

G = None
STEP_AUTORESET, STEP_ELIDE, STEP_TRUNCATE, STEP_RESAMPLE, STEP_DENSE, STEP_CONTINUE, STEP_ROS, STEP_FLAT_OBS, STEP_PACK_OBS = 1, 2, 4, 8, 16, 32, 64, 128, 256
AUG_PERMUTE, AUG_ROT90 = 1, 2


def golden():
    global G
    if G is None:
        z = np.load(os.path.join(B.GOLDEN_DIR, "research.npz"))
        G = {k: z[k] for k in z.files}
        G["traces"] = json.load(open(os.path.join(B.GOLDEN_DIR, "research_traces.json")))["replay_traces"]
    return G


def crop_table():
    ops = O.o2arc_ops()
    ops[33] = O.desc(O.OP_CROP_GRID, 0, O.F_RESET_SEL)  # agents/env.py:23-28
    return ops


def _state_fields(be, kind="o2arc"):
    return [f for f in O.PLANES[:-1] if f in O.KIND_PLANES[kind]] + [f for f in O.REC if f != "answer_dim"]


def wrappers(cls):
    """bbox / point ingress == the masks BBoxWrapper.action / PointWrapper.action of the reference build."""
    g, errs = golden(), []
    for H, W in ((30, 30), (7, 12)):
        k = f"{H}x{W}"
        for ing, tup, msk in (("bbox", g[f"wrap_bbox_{k}"], g[f"wrap_bbox_mask_{k}"]), ("point", g[f"wrap_point_{k}"], g[f"wrap_point_mask_{k}"])):
            N = len(tup)
            rng = np.random.default_rng(5)
            inp = rng.integers(0, 10, (N, H, W)).astype(np.int8)
            dims = np.tile(np.array([[H, W]], np.int8), (N, 1))
            a, b = cls(N, H, W, -1, "o2arc", O.o2arc_ops()), cls(N, H, W, -1, "o2arc", O.o2arc_ops())
            for be in (a, b):
                be.set_tasks(inp, dims, inp, dims)
                be.reset()
            op = (tup[:, -1] % 35).astype(np.int32)
            op[op >= 24] = 20 + op[op >= 24] % 4  # keep Rotate out (7x12 is not square)
            a.step(ing, tup[:, :-1], op)
            b.step("mask", msk, op)
            for f in _state_fields(a):
                if not np.array_equal(a.get(f), b.get(f)):
                    errs.append(f"{k} {ing}: field {f} differs between tuple ingress and the reference wrapper's mask")
    return errs


def augment(cls):
    """reset with colour permutation + rot90 == CustomO2ARCEnv.reset (agents/env.py:31-42)."""
    g, errs = golden(), []
    N, H, W = g["aug_in"].shape
    be = cls(N, H, W, -1, "o2arc", O.o2arc_ops())
    ins = [g["aug_in"][n][:g["aug_in_dim"][n, 0], :g["aug_in_dim"][n, 1]] for n in range(N)]
    outs = [g["aug_ans"][n][:g["aug_ans_dim"][n, 0], :g["aug_ans_dim"][n, 1]] for n in range(N)]
    be.set_task_table(ins, outs)
    be.reset_from_table(np.arange(N), None, g["aug_k"], g["aug_perm"])
    for f, want in (("input", g["aug_out_in"]), ("grid", g["aug_out_in"]), ("answer", g["aug_out_ans"]),
                    ("input_dim", g["aug_out_in_dim"]), ("grid_dim", g["aug_out_in_dim"]), ("answer_dim", g["aug_out_ans_dim"])):
        if not np.array_equal(be.get(f), want):
            bad = np.nonzero((be.get(f) != want).reshape(N, -1).any(1))[0]
            errs.append(f"augmented reset: {f} differs for envs {bad.tolist()} (k {g['aug_k'][bad].tolist()})")
    if be.status():
        errs.append("augmented reset raised a status flag")
    return errs


def dense(cls):
    """ARCLE_STEP_DENSE -> sparse*100 - 1 + correct/total == CustomO2ARCEnv.reward (agents/env.py:44-58), float64-exact."""
    g, errs = golden(), []
    S, N, H, W = g["dense_mask"].shape
    be = cls(N, H, W, -1, "o2arc", crop_table())
    be.set_tasks(g["aug_out_in"], g["aug_out_in_dim"], g["aug_out_ans"], g["aug_out_ans_dim"])
    be.reset()
    be.set_dense_output()
    for s in range(S):
        r, t = be.step("mask", g["dense_mask"][s], g["dense_op"][s], STEP_DENSE)
        d = be.dense.astype(np.float64)
        got = r.astype(np.float64) * 100 - 1 + d[:, 0] / d[:, 1]
        if not np.array_equal(got, g["dense_reward"][s]):
            bad = np.nonzero(got != g["dense_reward"][s])[0]
            errs.append(f"dense reward step {s}: envs {bad.tolist()} got {got[bad].tolist()} want {g['dense_reward'][s][bad].tolist()}")
        if not np.array_equal(t, g["dense_term"][s]):
            errs.append(f"dense step {s}: terminated differs")
        if len(errs) > 8:
            break
    if not np.array_equal(be.get("grid"), g["dense_final_grid"]) or not np.array_equal(be.get("grid_dim"), g["dense_final_grid_dim"]):
        errs.append("dense: final grid differs")
    return errs


def reset_on_submit(cls):
    """ARCLE_STEP_RESET_ON_SUBMIT == reset(options={'reset_on_submit': True}) of the reference (base.py:179-180)."""
    g, errs = golden(), []
    S, N, H, W = g["ros_mask"].shape
    for mt in sorted(set(g["ros_max_trial"].tolist())):
        sel = np.nonzero(g["ros_max_trial"] == mt)[0]
        be = cls(len(sel), H, W, int(mt), "o2arc", O.o2arc_ops())
        be.set_tasks(g["ros_in"][sel], g["ros_in_dim"][sel], g["ros_ans"][sel], g["ros_ans_dim"][sel])
        be.reset()
        for s in range(S):
            r, t = be.step("mask", g["ros_mask"][s][sel], g["ros_op"][s][sel], STEP_ROS)
            cnt = be.counters()
            checks = [("reward", r, g["ros_reward"][s][sel]), ("term", t, g["ros_term"][s][sel]),
                      ("steps", cnt[:, 0], g["ros_steps"][s][sel]), ("submit_count", cnt[:, 1], g["ros_submit"][s][sel])]
            checks += [(f, be.get(f), g["ros_" + f][s][sel]) for f in ("grid", "grid_dim", "selected", "clip", "trials_remain", "terminated")]
            for name, got, want in checks:
                if not np.array_equal(np.asarray(got).reshape(len(sel), -1), np.asarray(want).reshape(len(sel), -1)):
                    errs.append(f"reset_on_submit max_trial {mt} step {s}: {name} differs (ops {g['ros_op'][s][sel].tolist()})")
            if len(errs) > 8:
                return errs
    return errs


def flat(cls):
    """flattened observation rows (full / FilterO2ARC) == gymnasium FlattenObservation of the reference's state."""
    g, errs = golden(), []
    S, N, H, W = g["flat_mask"].shape
    be = cls(N, H, W, 5, "o2arc", O.o2arc_ops())
    be.set_tasks(g["flat_in"], g["flat_in_dim"], g["flat_ans"], g["flat_ans_dim"])
    be.reset()
    for s in range(S):
        be.step("mask", g["flat_mask"][s], g["flat_op"][s])
    for filtered, want in ((False, g["flat_rows"]), (True, g["flat_rows_filtered"])):
        got = be.flat_obs(filtered)
        if got.shape != want.shape or not np.array_equal(got, want):
            bad = np.nonzero((got != want).any(1))[0] if got.shape == want.shape else "shape"
            errs.append(f"flat obs (filtered={filtered}): rows differ for envs {bad if isinstance(bad, str) else bad.tolist()}")
    # the same rows written by the step kernel itself (STEP_FLAT_OBS, fused epilogue): replay the trace with the flag set; the
    # rows after the last step are the golden ones, and after every step they equal the stand-alone writer's
    for filtered, want in ((False, g["flat_rows"]), (True, g["flat_rows_filtered"])):
        be = cls(N, H, W, 5, "o2arc", O.o2arc_ops())
        be.set_tasks(g["flat_in"], g["flat_in_dim"], g["flat_ans"], g["flat_ans_dim"])
        be.reset()
        be.set_flat_output(filtered)
        for s in range(S):
            be.step("mask", g["flat_mask"][s], g["flat_op"][s], STEP_FLAT_OBS)
            if s % 5 == 0 and not np.array_equal(be.fused_flat(), be.flat_obs(filtered)):
                errs.append(f"fused flat obs (filtered={filtered}) differs from the stand-alone writer after step {s}")
        got = be.fused_flat()
        if got.shape != want.shape or not np.array_equal(got, want):
            errs.append(f"fused flat obs (filtered={filtered}): rows differ from the golden rows")
    return errs


def continue_rule(cls):
    """ARCLE_STEP_CONTINUE_RULE on the LOGGED selections == the reference harness's rule (tests/o2arc_check.py:169-170):
    grids after every step of the synthetic O2ARC traces."""
    g, errs = golden(), []
    n, T = g["replay_op"].shape
    be = cls(n, 30, 30, -1, "o2arc", O.o2arc_ops())
    be.set_tasks(g["replay_in"], g["replay_in_dim"], g["replay_ans"], g["replay_ans_dim"])
    be.reset()
    for t in range(T):
        live = g["replay_op"][:, t] >= 0
        op = np.where(live, g["replay_op"][:, t], 32).astype(np.int32)  # finished traces: any op, no longer compared
        be.step("mask", g["replay_sel"][:, t], op, STEP_CONTINUE)
        grid, dim = be.get("grid"), be.get("grid_dim")
        bad = [i for i in np.nonzero(live)[0] if not (np.array_equal(grid[i], g["replay_grid"][i, t]) and np.array_equal(dim[i], g["replay_grid_dim"][i, t]))]
        if bad:
            errs.append(f"trace replay step {t}: grid differs for traces {bad} (ops {op[bad].tolist()})")
            break
    return errs


def sampler(cls, kind_flags=AUG_PERMUTE | AUG_ROT90):
    """Device task draws: (1) equal the host mirror arcle_amd.sampling.draw_task, (2) depend on the GLOBAL env id only —
    two half shards reproduce the whole batch exactly, through sampled resets and ARCLE_STEP_RESAMPLE episodes."""
    from arcle_amd.sampling import draw_task
    errs = []
    H = W = 12
    N, T, S, seed = 16, 9, 30, 0xC0FFEE12345
    rng = np.random.default_rng(3)
    ins = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
    outs = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
    pair_off, pair_cnt = np.array([0, 2, 3, 7], np.int32), np.array([2, 1, 4, 2], np.int32)  # 4 problems over the 9 entries

    def make(n, base):
        be = cls(n, H, W, 2, "o2arc", O.o2arc_ops())
        be.set_task_table(ins, outs)
        be.set_sampler(pair_off, pair_cnt, seed, base, kind_flags)
        be.set_truncation(7)
        be.reset_sampled()
        return be

    whole, lo, hi = make(N, 0), make(N // 2, 0), make(N // 2, N // 2)
    want = [pair_off[p] + s for p, s, _, _ in (draw_task(seed, gid, 0, pair_cnt, kind_flags) for gid in range(N))]
    if whole.cur_task.tolist() != [int(x) for x in want]:
        errs.append(f"first draw: device {whole.cur_task.tolist()} != host mirror {want}")
    bb = rng.integers(0, H, (S, N, 4)).astype(np.int32)
    op = rng.choice(35, (S, N), p=np.r_[[1] * 34, [12]] / 46.0).astype(np.int32)  # many submits: episodes end (max_trial 2)
    fl = STEP_RESAMPLE | STEP_TRUNCATE
    for s in range(S):
        whole.step("bbox", bb[s], op[s], fl)
        lo.step("bbox", bb[s, :N // 2], op[s, :N // 2], fl)
        hi.step("bbox", bb[s, N // 2:], op[s, N // 2:], fl)
        for f in ["input", "answer", "grid", "selected", "input_dim", "grid_dim", "answer_dim", "trials_remain", "terminated"]:
            if not np.array_equal(whole.get(f), np.concatenate([lo.get(f), hi.get(f)])):
                errs.append(f"step {s}: field {f} of the sharded run differs from the whole batch")
        if not np.array_equal(whole.episode, np.concatenate([lo.episode, hi.episode])):
            errs.append(f"step {s}: episode counters differ")
        if len(errs) > 6:
            break
    ep = whole.episode
    if ep.max() < 3:
        errs.append(f"the trace did not exercise re-sampling (episodes {ep.tolist()})")
    want = [pair_off[p] + s for p, s, _, _ in (draw_task(seed, gid, int(ep[gid]) - 1, pair_cnt, kind_flags) for gid in range(N))]
    if whole.cur_task.tolist() != [int(x) for x in want]:
        errs.append("current task after re-sampling differs from the host mirror")
    if whole.status() | lo.status() | hi.status():
        errs.append("status flag raised")
    return errs


def resample_content(cls):
    """What a device-drawn, augmented task LOOKS like after sampled resets and ARCLE_STEP_RESAMPLE auto-resets inside the step kernel
    (where the wave runs the width class's fast paths: v_perm colour lookup, affine window gather for np.rot90): input / answer planes
    and dims of every env == np.rot90(perm[pair], k) of the table entry the host mirror predicts for (seed, global env id, episode) —
    30 x 30 (FW_FULL), 20 x 24 (FW_FAST, non-square: a quarter turn that does not fit is dropped), 12 x 12 (generic width)."""
    from arcle_amd.sampling import draw_task
    errs = []
    for (H, W), seed in (((30, 30), 11), ((20, 24), 12), ((12, 12), 13)):
        N, T, S = 24, 12, 14
        rng = np.random.default_rng(seed)
        ins = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
        outs = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
        pair_off, pair_cnt = np.array([0, 3, 4, 9], np.int32), np.array([3, 1, 5, 3], np.int32)
        be = cls(N, H, W, 1, "o2arc", O.o2arc_ops())
        be.set_task_table(ins, outs)
        be.set_sampler(pair_off, pair_cnt, 1000 + seed, 7, AUG_PERMUTE | AUG_ROT90)
        be.set_truncation(3)
        be.reset_sampled()

        def check(tag):
            ep, got_in, got_an = be.episode, be.get("input"), be.get("answer")
            din, dan = be.get("input_dim"), be.get("answer_dim")
            for n in range(N):
                p_, s_, k, perm = draw_task(1000 + seed, 7 + n, int(ep[n]) - 1, pair_cnt, AUG_PERMUTE | AUG_ROT90)
                t = pair_off[p_] + s_
                a, b = ins[t], outs[t]
                if k & 1 and (a.shape[1] > H or a.shape[0] > W or b.shape[1] > H or b.shape[0] > W):
                    k &= 2  # (documented softening of a quarter turn that does not fit a non-square plane)
                lut = np.asarray(perm, np.int8)
                wa, wb = np.rot90(lut[a], k), np.rot90(lut[b], k)
                ea, eb = np.zeros((H, W), np.int8), np.zeros((H, W), np.int8)
                ea[:wa.shape[0], :wa.shape[1]] = wa
                eb[:wb.shape[0], :wb.shape[1]] = wb
                if not (np.array_equal(got_in[n], ea) and np.array_equal(got_an[n], eb) and tuple(din[n]) == wa.shape and tuple(dan[n]) == wb.shape):
                    errs.append(f"{H}x{W} {tag}: env {n} (episode {int(ep[n])}, entry {t}, k {k}) does not hold the augmented task the mirror predicts")
        check("after reset_sampled")
        bb = rng.integers(0, H, (S, N, 4)).astype(np.int32)
        bb[..., 1], bb[..., 3] = bb[..., 1] % W, bb[..., 3] % W
        for s in range(S):
            be.step("bbox", bb[s], rng.integers(0, 35, N).astype(np.int32), STEP_RESAMPLE | STEP_TRUNCATE | STEP_ELIDE)
            if s % 4 == 3:
                check(f"step {s}")
            if len(errs) > 6:
                return errs
        if be.episode.min() < 3:
            errs.append(f"{H}x{W}: the trace did not exercise re-sampling")
        if be.status():
            errs.append(f"{H}x{W}: status flag raised")
    return errs


def truncation(cls):
    errs = []
    N, H, W = 8, 10, 10
    be = cls(N, H, W, -1, "o2arc", O.o2arc_ops())
    inp = np.ones((N, H, W), np.int8)
    dims = np.tile(np.array([[H, W]], np.int8), (N, 1))
    be.set_tasks(inp, dims, inp * 2, dims)
    be.reset()
    be.set_truncation(5)
    bb = np.zeros((N, 4), np.int32)
    for s in range(1, 9):
        be.step("bbox", bb, np.full(N, 3, np.int32), STEP_TRUNCATE)
        if not np.array_equal(be.trunc, np.full(N, s >= 5, np.uint8)):
            errs.append(f"step {s}: truncated {be.trunc.tolist()}")
    # with autoreset the truncated episode restarts on the next step
    be.step("bbox", bb, np.full(N, 3, np.int32), STEP_TRUNCATE | STEP_AUTORESET)
    if be.counters()[:, 0].tolist() != [0] * N:
        errs.append("autoreset did not restart the truncated episodes")
    return errs


def packed(cls):
    errs = []
    for H, W in ((30, 30), (10, 10), (5, 7)):
        N = 12
        rng = np.random.default_rng(H)
        be = cls(N, H, W, -1, "o2arc", O.o2arc_ops())
        inp = rng.integers(0, 10, (N, H, W)).astype(np.int8)
        dims = np.stack([rng.integers(1, H + 1, N), rng.integers(1, W + 1, N)], 1).astype(np.int8)
        be.set_tasks(inp, dims, inp, dims)
        be.reset()
        r, t = be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 34, np.int32))
        rows = be.packed_obs()
        P = H * W
        want = np.concatenate([be.get("grid").reshape(N, P).view(np.uint8), be.get("grid_dim").view(np.uint8),
                               r.astype("<i4").view(np.uint8).reshape(N, 4), t.reshape(N, 1)], 1)
        if rows.shape[1] != ((P + 7 + 15) & ~15) or not np.array_equal(rows[:, :P + 7], want) or rows[:, P + 7:].any():
            errs.append(f"{H}x{W}: packed observation rows differ")
        # the same rows written by the step kernel itself (STEP_PACK_OBS, fused epilogue) over a short random trace
        be.set_packed_output()
        for s_ in range(6):
            bb = np.stack([rng.integers(0, H, N), rng.integers(0, W, N), rng.integers(0, H, N), rng.integers(0, W, N)], 1).astype(np.int32)
            op = rng.integers(0, 35, N).astype(np.int32)
            r, t = be.step("bbox", bb, op, STEP_AUTORESET | STEP_PACK_OBS)
            fused = be.fused_packed()
            want = np.concatenate([be.get("grid").reshape(N, P).view(np.uint8), be.get("grid_dim").view(np.uint8),
                                   r.astype("<i4").view(np.uint8).reshape(N, 4), t.reshape(N, 1)], 1)
            if not np.array_equal(fused[:, :P + 7], want) or fused[:, P + 7:].any() or not np.array_equal(fused, be.packed_obs()):
                errs.append(f"{H}x{W}: fused packed rows differ at step {s_}")
                break
    return errs


def bad_selection(cls):
    """A point outside the plane / negative coordinates: empty selection + ARCLE_ST_BAD_SELECTION, on the backend and the
    oracle alike (the reference's wrappers raise IndexError or wrap the index)."""
    errs = []
    N, H, W = 4, 10, 10
    ops = O.o2arc_ops()
    be, orc = cls(N, H, W, -1, "o2arc", ops), B.OracleBackend(N, H, W, -1, "o2arc", ops)
    inp = np.ones((N, H, W), np.int8)
    dims = np.tile(np.array([[H, W]], np.int8), (N, 1))
    for b in (be, orc):
        b.set_tasks(inp, dims, inp, dims)
        b.reset()
    for ing, pay in (("point", np.array([[H, 0], [0, W], [-1, 2], [3, 3]], np.int32)),
                     ("bbox", np.array([[-1, 0, 2, 2], [0, -3, 1, 1], [1, 1, 2, 2], [H, W, H, W]], np.int32))):
        op = np.full(N, 4, np.int32)
        be.step(ing, pay, op)
        orc.step(ing, pay, op)
        s1, s2 = be.status(), orc.status()
        if s1 != 8 or s2 != 8:
            errs.append(f"{ing}: status {s1} (oracle {s2}), expected ARCLE_ST_BAD_SELECTION")
        if not np.array_equal(be.get("grid"), orc.get("grid")):
            errs.append(f"{ing}: grid differs from the oracle")
    return errs
