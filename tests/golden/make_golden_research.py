#!/usr/bin/env python3
"""Golden vectors for the callers / data formats either side of the hot path, captured from the
UNMODIFIED reference (/root/reference, imported through oracle/stubs).  Build-container only; the resulting
tests/golden/research_*.npz / .json are pure data.

    python tests/golden/make_golden_research.py

  wrappers   BBoxWrapper.action / PointWrapper.action (arcle/wrappers/bbox.py:22-30,43-49): tuples -> masks
  augment    CustomO2ARCEnv.reset (agents/env.py:31-42): colour permutation + rot90 of (input, answer)
  dense      CustomO2ARCEnv.reward (agents/env.py:44-58) along random traces on its 35-op table (op 33 = crop)
  ros        O2ARCv2Env with reset(options={'reset_on_submit': True}) (base.py:179-180)
  flat       FlattenObservation rows of O2ARCv2Env states, full and through FilterO2ARC (agents/env.py:109-126); the full
             layout is cross-checked with the reference's own unflatten_vec (agents/models/GPTPolicy.py:17-42)
  replay     synthetic O2ARC web-UI logs through the reference's action_convert (tests/o2arc_check.py:21-99, extracted
             from the script with `ast`, executed unmodified) and its continuation rule (:169-170)
"""
import ast
import importlib.util
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refdriver as RD  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = RD.REFERENCE_ROOT


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _one_task_loader(ti, to):
    from arcle.loaders import Loader

    class OneTask(Loader):
        def get_path(self, **kw):
            return [""]

        def parse(self, **kw):
            return [([ti], [to], [ti], [to], {"id": "synthetic"})]
    return OneTask()


def _pad(a, H, W):
    out = np.zeros((H, W), np.int8)
    out[:a.shape[0], :a.shape[1]] = a
    return out


def _random_action(rng, H, W, n_ops, heavy_submit=False):
    kind, payload, m = RD.random_selection(rng, H, W)
    if heavy_submit and rng.chance(1, 4):
        return n_ops - 1, m
    return RD.pick_op(rng, n_ops, "o2arc"), m


def wrappers(out):
    from arcle.envs import O2ARCv2Env
    from arcle.wrappers import BBoxWrapper, PointWrapper
    rng = RD.SplitMix64(301)
    for H, W in ((30, 30), (7, 12)):
        env = O2ARCv2Env(data_loader=_one_task_loader(np.ones((2, 2), np.int8), np.ones((2, 2), np.int8)), max_grid_size=(H, W))
        bw, pw = BBoxWrapper(env), PointWrapper(env)
        tb = np.array([[rng.below(H), rng.below(W), rng.below(H), rng.below(W), rng.below(35)] for _ in range(48)], np.int32)
        tp = np.array([[rng.below(H), rng.below(W), rng.below(35)] for _ in range(24)], np.int32)
        mb = np.stack([bw.action(tuple(int(v) for v in t))["selection"] for t in tb])
        mp = np.stack([pw.action(tuple(int(v) for v in t))["selection"] for t in tp])
        assert mb.dtype == np.int8 and mp.dtype == np.int8
        k = f"{H}x{W}"
        out[f"wrap_bbox_{k}"], out[f"wrap_bbox_mask_{k}"] = tb, mb
        out[f"wrap_point_{k}"], out[f"wrap_point_mask_{k}"] = tp, mp


def augment_and_dense(out):
    renv = _load("ref_agents_env", os.path.join(REF, "agents", "env.py"))
    H = W = 12
    N, S = 12, 48
    rng = RD.SplitMix64(302)
    tin, tan, tind, tand, ks, perms = [], [], [], [], [], []
    oin, oan, oind, oand = [], [], [], []
    envs = []
    for n in range(N):
        ti, to = RD.random_task(rng, H - 2, W - 3)
        env = renv.CustomO2ARCEnv(data_loader=_one_task_loader(ti, to), max_grid_size=(H, W), max_trial=-1)
        env.reset_options = {"adaptation": True, "prob_index": 0}
        seed = 1000 + n
        np.random.seed(seed)
        env.reset()
        np.random.seed(seed)  # replay the draws of base.py:99 (subprob), agents/env.py:33-34 (k, permutation)
        np.random.randint(0, 1)
        k = int(np.random.randint(0, 4))
        perm = np.random.permutation(10)
        tin.append(_pad(ti, H, W)); tan.append(_pad(to, H, W)); tind.append(ti.shape); tand.append(to.shape)
        ks.append(k); perms.append(perm)
        st = env.current_state
        assert np.array_equal(st["input"], _pad(env.input_, H, W)) and np.array_equal(st["grid"], st["input"])
        oin.append(st["input"].copy()); oind.append(env.input_.shape)
        oan.append(_pad(env.answer, H, W)); oand.append(env.answer.shape)
        envs.append(env)
    out.update(aug_in=np.stack(tin), aug_in_dim=np.array(tind, np.int8), aug_ans=np.stack(tan),
               aug_ans_dim=np.array(tand, np.int8), aug_k=np.array(ks, np.uint8), aug_perm=np.array(perms, np.uint8),
               aug_out_in=np.stack(oin), aug_out_in_dim=np.array(oind, np.int8), aug_out_ans=np.stack(oan),
               aug_out_ans_dim=np.array(oand, np.int8))
    # dense reward along random traces, starting from the augmented tasks
    n_ops = len(envs[0].operations)
    ops = np.zeros((S, N), np.int32)
    masks = np.zeros((S, N, H, W), np.int8)
    rew = np.zeros((S, N), np.float64)
    term = np.zeros((S, N), np.uint8)
    for s in range(S):
        for n, env in enumerate(envs):
            o, m = _random_action(rng, H, W, n_ops, heavy_submit=True)
            ops[s, n], masks[s, n] = o, m
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                _, r, t, _, _ = env.step({"selection": m.copy(), "operation": int(o)})
            rew[s, n], term[s, n] = float(r), int(bool(t))
    out.update(dense_op=ops, dense_mask=masks, dense_reward=rew, dense_term=term,
               dense_final_grid=np.stack([e.current_state["grid"] for e in envs]),
               dense_final_grid_dim=np.stack([e.current_state["grid_dim"] for e in envs]))


def reset_on_submit(out):
    from arcle.envs import O2ARCv2Env
    H = W = 10
    N, S = 8, 64
    rng = RD.SplitMix64(303)
    tasks = [RD.random_task(rng, H, W) for _ in range(N)]
    envs = []
    for n, (ti, to) in enumerate(tasks):
        env = O2ARCv2Env(data_loader=_one_task_loader(ti, to), max_grid_size=(H, W), max_trial=[-1, 3, 1, 0][n % 4])
        env.reset(options={"prob_index": 0, "subprob_index": 0, "reset_on_submit": True})
        envs.append(env)
    fields = ("grid", "grid_dim", "selected", "clip", "trials_remain", "terminated")
    ops = np.zeros((S, N), np.int32)
    masks = np.zeros((S, N, H, W), np.int8)
    rec = {f: [] for f in fields}
    rew = np.zeros((S, N), np.int32); term = np.zeros((S, N), np.uint8)
    steps = np.zeros((S, N), np.int32); subs = np.zeros((S, N), np.int32)
    for s in range(S):
        snap = {f: [] for f in fields}
        for n, env in enumerate(envs):
            o, m = _random_action(rng, H, W, 35, heavy_submit=True)
            ops[s, n], masks[s, n] = o, m
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                st, r, t, _, info = env.step({"selection": m.copy(), "operation": int(o)})
            rew[s, n], term[s, n], steps[s, n], subs[s, n] = int(r), int(bool(t)), info["steps"], info["submit_count"]
            for f in fields:
                snap[f].append(np.asarray(st[f], np.int8).copy())
        for f in fields:
            rec[f].append(np.stack(snap[f]))
    out.update(ros_in=np.stack([_pad(a, H, W) for a, _ in tasks]), ros_in_dim=np.array([a.shape for a, _ in tasks], np.int8),
               ros_ans=np.stack([_pad(b, H, W) for _, b in tasks]), ros_ans_dim=np.array([b.shape for _, b in tasks], np.int8),
               ros_max_trial=np.array([[-1, 3, 1, 0][n % 4] for n in range(N)], np.int32),
               ros_op=ops, ros_mask=masks, ros_reward=rew, ros_term=term, ros_steps=steps, ros_submit=subs)
    for f in fields:
        out["ros_" + f] = np.stack(rec[f])


def flat(out):
    import gymnasium
    import torch
    from arcle.envs import O2ARCv2Env
    renv = _load("ref_agents_env2", os.path.join(REF, "agents", "env.py"))
    gpt = _load("ref_gptpolicy", os.path.join(REF, "agents", "models", "GPTPolicy.py"))
    H = W = 12
    N, S = 10, 40
    rng = RD.SplitMix64(304)
    tasks = [RD.random_task(rng, H, W) for _ in range(N)]
    ops = np.zeros((S, N), np.int32)
    masks = np.zeros((S, N, H, W), np.int8)
    rows, frows = [], []
    for n, (ti, to) in enumerate(tasks):
        env = O2ARCv2Env(data_loader=_one_task_loader(ti, to), max_grid_size=(H, W), max_trial=5)
        fenv = renv.FilterO2ARC(env)
        env.reset(options={"prob_index": 0, "subprob_index": 0})
        for s in range(S):
            o, m = _random_action(rng, H, W, 35)
            ops[s, n], masks[s, n] = o, m
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                st, *_ = env.step({"selection": m.copy(), "operation": int(o)})
        rows.append(gymnasium.spaces.flatten(env.observation_space, st).astype(np.int8))
        frows.append(gymnasium.spaces.flatten(fenv.observation_space, fenv.observation(st)).astype(np.int8))
        # the reference's own consumer of the full layout
        un = gpt.unflatten_vec(torch.from_numpy(rows[-1].astype(np.int64))[None], H)
        flat_state = RD.flatten_state(st)
        for k, v in un.items():
            assert np.array_equal(v[0].numpy().reshape(-1), np.asarray(flat_state[k], np.int64).reshape(-1)), k
    out.update(flat_in=np.stack([_pad(a, H, W) for a, _ in tasks]), flat_in_dim=np.array([a.shape for a, _ in tasks], np.int8),
               flat_ans=np.stack([_pad(b, H, W) for _, b in tasks]), flat_ans_dim=np.array([b.shape for _, b in tasks], np.int8),
               flat_op=ops, flat_mask=masks, flat_rows=np.stack(rows), flat_rows_filtered=np.stack(frows))


def _reference_action_convert():
    """The function `action_convert` of the reference's trace harness, compiled from its own source text (the script around
    it reads pickles that are not in the checkout, so the module cannot be imported)."""
    src = open(os.path.join(REF, "tests", "o2arc_check.py")).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "action_convert")
    ns = {"np": np}
    exec(compile(ast.Module([node], []), "o2arc_check.py:action_convert", "exec"), ns)
    return ns["action_convert"]


def replay(out, meta):
    from arcle.envs import O2ARCv2Env
    convert = _reference_action_convert()
    H = W = 30
    rng = RD.SplitMix64(305)
    traces, exp_ops, exp_sel, exp_grid, exp_dim, tasks = [], [], [], [], [], []
    for tr in range(10):
        ti, to = RD.random_task(rng, 12, 12)
        tasks.append((ti, to))
        env = O2ARCv2Env(data_loader=_one_task_loader(ti, to), max_grid_size=(H, W))
        obs, _ = env.reset(options={"adaptation": False, "prob_index": 0, "subprob_index": 0})
        entries, e_ops, e_sel, e_grid, e_dim = [], [], [], [], []
        for i in range(24):
            gh, gw = int(obs["grid_dim"][0]), int(obs["grid_dim"][1])
            h0, w0 = rng.below(gh), rng.below(gw)
            h1, w1 = min(gh - 1, h0 + rng.below(4)), min(gw - 1, w0 + rng.below(4))
            t = rng.below(100)
            prev = entries[-1] if entries else None
            if prev and prev[0] in ("Move", "RotateCW", "RotateCCW", "FlipX", "FlipY") and rng.chance(2, 3) and obs["selected"].any():
                # the web UI logs the selection at the object's CURRENT place: the bbox of obs['selected']
                rows = np.flatnonzero(obs["selected"].any(1)); cols = np.flatnonzero(obs["selected"].any(0))
                box = [[int(rows[0]), int(cols[0])], [int(rows[-1]), int(cols[-1])]]
                name = ["Move", "RotateCW", "RotateCCW", "FlipX", "FlipY"][rng.below(5)]
                data = box + (["UDRL"[rng.below(4)]] if name == "Move" else [])
            elif t < 12:
                name, data = "Color", [[h0, w0], rng.below(10)]
            elif t < 24:
                name, data = "Fill", [[h0, w0], [h1, w1], rng.below(10)]
            elif t < 44:
                name, data = "Move", [[h0, w0], [h1, w1], "UDRL"[rng.below(4)]]
            elif t < 54:
                name, data = ["RotateCW", "RotateCCW", "FlipX", "FlipY"][rng.below(4)], [[h0, w0], [h1, w1]]
            elif t < 62:
                name, data = "Copy", [[h0, w0], [h1, w1], ["Input Grid", "Output Grid"][rng.below(2)]]
            elif t < 68:
                name, data = "Paste", [[h0, w0]]
            elif t < 78:
                name, data = "FloodFill", [[h0, w0], rng.below(10)]
            elif t < 84:
                name, data = "ResizeGrid", [[1 + rng.below(12), 1 + rng.below(12)]]
            elif t < 90:
                name, data = "CopyFromInput", []
            elif t < 94:
                name, data = "ResetGrid", []
            else:
                name, data = "Submit", []
            op, sel = convert((None, name, [tuple(d) if isinstance(d, list) else d for d in data], None))
            sent = np.zeros((30, 30), np.bool_) if (20 <= op <= 27 and np.all(obs["selected"] == sel)) else sel  # :169-170
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                obs, r, term, trunc, info = env.step({"selection": sent, "operation": op})
            entries.append([name, data])
            e_ops.append(op); e_sel.append(sel.astype(np.int8)); e_grid.append(obs["grid"].copy()); e_dim.append(obs["grid_dim"].copy())
            if term or trunc:
                break
        traces.append(entries)
        exp_ops.append(e_ops); exp_sel.append(e_sel); exp_grid.append(e_grid); exp_dim.append(e_dim)
    T = max(len(t) for t in traces)
    n = len(traces)
    ops = np.full((n, T), -1, np.int32); sel = np.zeros((n, T, 30, 30), np.int8)
    grid = np.zeros((n, T, 30, 30), np.int8); dim = np.zeros((n, T, 2), np.int8)
    for i in range(n):
        L = len(traces[i])
        ops[i, :L] = exp_ops[i]; sel[i, :L] = np.stack(exp_sel[i]); grid[i, :L] = np.stack(exp_grid[i]); dim[i, :L] = np.stack(exp_dim[i])
    out.update(replay_in=np.stack([_pad(a, 30, 30) for a, _ in tasks]), replay_in_dim=np.array([a.shape for a, _ in tasks], np.int8),
               replay_ans=np.stack([_pad(b, 30, 30) for _, b in tasks]), replay_ans_dim=np.array([b.shape for _, b in tasks], np.int8),
               replay_op=ops, replay_sel=sel, replay_grid=grid, replay_grid_dim=dim)
    meta["replay_traces"] = traces


def main():
    RD.import_reference()
    out, meta = {}, {}
    wrappers(out)
    augment_and_dense(out)
    reset_on_submit(out)
    flat(out)
    replay(out, meta)
    np.savez_compressed(os.path.join(OUT, "research.npz"), **out)
    json.dump(meta, open(os.path.join(OUT, "research_traces.json"), "w"))
    print("research.npz:", {k: v.shape for k, v in out.items()})
    print(f"{os.path.getsize(os.path.join(OUT, 'research.npz')) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
