#!/usr/bin/env python3
"""Captures golden vectors from the UNMODIFIED reference (/root/reference, imported through the
stubs in oracle/stubs).  Build-container only; the resulting tests/golden/*.npz are committed and
are pure data (tasks, actions, expected states/hashes) — no reference source travels.

    python tests/golden/make_golden.py            # rewrites every fixture

Fixture layout (one .npz per configuration; N traces stepped as N independent envs, S steps):
  meta            json string: variant, kind, H, W, max_trial, N, S, ops (descriptor table)
  input, answer   int8 [N,H,W] zero-padded task grids; input_dim, answer_dim int8 [N,2]
  ingress         uint8 [S]     0 = bbox, 1 = point, 2 = mask  (same form for all envs of a step)
  op              int32 [S,N]
  bbox            int32 [S,N,4] (x1,y1,x2,y2), valid where ingress==0
  xy              int32 [S,N,2] valid where ingress==1
  mask_steps      int32 [M]     the step indices with ingress==2;  masks int8 [M,N,H,W]
  reward          int32 [S,N]; term uint8 [S,N]; steps, submit_count int32 [S,N]
  hash            uint64 [S,N,F] position-weighted checksum (see `checksum`) of every state field
                  after every step, F = len(fields); `fields` is a json list of field names
  full_steps      int32 [K] steps after which the complete state is stored;
  full_<field>    int8 [K,N,...] the complete state at those steps (always includes the last step)
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refdriver as RD  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PLANE_FIELDS = ["input", "grid", "selected", "clip", "object", "object_sel", "background"]
SCALAR_FIELDS = ["input_dim", "grid_dim", "clip_dim", "object_dim", "object_pos", "trials_remain",
                 "terminated", "active", "rotation_parity"]


def weights(n):
    """Fixed odd 64-bit multipliers (splitmix64 stream) for the checksum."""
    r = RD.SplitMix64(0xC0FFEE)
    return np.array([r.next() | 1 for _ in range(n)], np.uint64)


_W = weights(16384)  # (a prefix of the same stream: fixtures of at most 1024 cells are unchanged)


def checksum(a):
    """a: int8 [..., L] -> uint64 [...]: sum_k (uint8(a_k) + 1) * w_k  mod 2^64 (position sensitive)."""
    a = np.asarray(a)
    flat = a.reshape(a.shape[0], -1).view(np.uint8).astype(np.uint64) + np.uint64(1)
    with np.errstate(over="ignore"):
        return (flat * _W[: flat.shape[1]]).sum(axis=1, dtype=np.uint64)


def fields_of(variant):
    kind, _ = RD.variant_table(variant)
    if kind == "o2arc":
        return PLANE_FIELDS + SCALAR_FIELDS
    if kind == "arc":
        return ["input", "grid", "clip", "input_dim", "grid_dim", "clip_dim", "trials_remain", "terminated"]
    return ["input", "grid", "input_dim", "grid_dim", "trials_remain", "terminated"]


def capture(name, variant, H, W, max_trial, N, S, seed, script=None, weird=False, full_every=16, bool_masks=False):
    """script(rng, s, n, H, W, n_ops) -> (op, mask) overrides the random action stream (ingress = mask)."""
    rng = RD.SplitMix64(seed)
    kind, table = RD.variant_table(variant)
    tasks = [RD.random_task(rng, H, W) for _ in range(N)]
    if script is not None and hasattr(script, "task"):
        tasks = [script.task(n, H, W) or tasks[n] for n in range(N)]
    envs = [RD.make_reference_env(variant, H, W, max_trial, t) for t in tasks]
    n_ops = len(table)
    fields = fields_of(variant)
    ingress = np.zeros(S, np.uint8)
    op = np.zeros((S, N), np.int32)
    bbox = np.zeros((S, N, 4), np.int32)
    xy = np.zeros((S, N, 2), np.int32)
    masks, mask_steps = [], []
    reward = np.zeros((S, N), np.int32)
    term = np.zeros((S, N), np.uint8)
    steps = np.zeros((S, N), np.int32)
    submit_count = np.zeros((S, N), np.int32)
    hashes = np.zeros((S, N, len(fields)), np.uint64)
    full_steps = [s for s in range(S) if (s + 1) % full_every == 0 or s == S - 1]
    full = {f: [] for f in fields}
    for s in range(S):
        ing = 2 if script is not None else [0, 0, 1, 2, 2][rng.below(5)]
        ingress[s] = ing
        step_masks = np.zeros((N, H, W), np.int8)
        snap = {f: [] for f in fields}
        for n, env in enumerate(envs):
            if script is not None:
                o, m = script(rng, s, n, H, W, n_ops)
            else:
                o = RD.pick_op(rng, n_ops, variant)
                while True:
                    sk, payload, m = RD.random_selection(rng, H, W, weird)
                    if (ing == 0 and sk == "bbox") or (ing == 1 and sk == "point") or (ing == 2 and sk == "mask"):
                        break
                if ing == 0:
                    bbox[s, n] = payload
                elif ing == 1:
                    xy[s, n] = payload
            op[s, n] = o
            step_masks[n] = m
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                # bool_masks: the reference is fed np.bool_ selections (base.py:136 allows both); the fixture stores 0/1
                _, r, t, trunc, info = env.step({"selection": m.astype(bool) if bool_masks else m.copy(),
                                                 "operation": int(o)})
            assert trunc is False
            reward[s, n], term[s, n] = int(r), int(bool(t))
            steps[s, n] = info["steps"]
            submit_count[s, n] = info.get("submit_count", env.submit_count)
            st = RD.flatten_state(env.current_state)
            for f in fields:
                assert st[f].dtype == np.int8, (f, st[f].dtype)
                snap[f].append(st[f].copy())
        if ing == 2:
            mask_steps.append(s)
            masks.append(step_masks)
        for fi, f in enumerate(fields):
            arr = np.stack(snap[f])
            hashes[s, :, fi] = checksum(arr)
            if s in full_steps:
                full[f].append(arr)
    meta = dict(variant=variant, kind=kind, H=H, W=W, max_trial=max_trial, N=N, S=S, bool_masks=bool_masks,
                ops=[int(x) for x in table], seed=seed)
    inp = np.zeros((N, H, W), np.int8)
    ans = np.zeros((N, H, W), np.int8)
    idim = np.zeros((N, 2), np.int8)
    adim = np.zeros((N, 2), np.int8)
    for n, (a, b) in enumerate(tasks):
        inp[n, :a.shape[0], :a.shape[1]] = a
        ans[n, :b.shape[0], :b.shape[1]] = b
        idim[n], adim[n] = a.shape, b.shape
    out = dict(meta=json.dumps(meta), fields=json.dumps(fields), input=inp, answer=ans, input_dim=idim,
               answer_dim=adim, ingress=ingress, op=op, bbox=bbox, xy=xy,
               mask_steps=np.asarray(mask_steps, np.int32),
               masks=np.stack(masks) if masks else np.zeros((0, N, H, W), np.int8),
               reward=reward, term=term, steps=steps, submit_count=submit_count, hash=hashes,
               full_steps=np.asarray(full_steps, np.int32))
    for f in fields:
        out["full_" + f] = np.stack(full[f])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {variant} {H}x{W} N={N} S={S} -> {os.path.getsize(path) / 1024:.0f} KiB")


# ---- scripted known-answer traces for the quirks of SURVEY.md A.6 -------------------------------
class QuirkScript:
    """Env n runs scenario n % 8; all are expressed as (op, mask) streams on the 35-op O2ARC table."""

    def task(self, n, H, W):
        rng = RD.SplitMix64(77 + n)
        h, w = (H, W) if n % 8 in (0, 1, 2, 3) else (max(1, H // 2), max(1, W - 3))
        a = np.array([[1 + rng.below(9) for _ in range(w)] for _ in range(h)], np.int8)
        return a, a.copy()

    def __call__(self, rng, s, n, H, W, n_ops):
        m = np.zeros((H, W), np.int8)
        sc = n % 8
        if sc in (0, 1, 2, 3):
            # A.6-5: object_pos wraps mod 256 — select a 2x3 block once, then Move in one direction forever
            if s == 0:
                m[3:5, 4:7] = 1
            return 20 + sc, m
        if sc == 4:
            # A.5: trials_remain int8 wrap (max_trial=-1): Submit every step
            return 34, m
        if sc == 5:
            # A.6-3/-4: Color and Paste write outside grid_dim; Copy bound check is off by one
            gh, gw = max(1, H // 2), max(1, W - 3)
            cyc = s % 6
            if cyc == 0:
                m[gh - 1:H, gw - 1:W] = 1          # Color into the padding
                return 3, m
            if cyc == 1:
                m[0:gh + 1, 0:gw + 1] = 1          # CopyO with xmax == gh (accepted, copies padding)
                return 29, m
            if cyc == 2:
                m[H - 2, W - 2] = 1                # Paste clipped at HxW, not grid_dim
                return 30, m
            if cyc == 3:
                m[0:gh + 2, 0:2] = 1               # CopyO with xmax == gh+1 (rejected)
                return 29, m
            if cyc == 4:
                m[gh - 1:gh + 1, :] = 1            # lift an object that includes padding cells
                return 21, m
            return 33 if s % 12 == 5 else 26, m    # ResizeGrid (empty sel: no-op) / FlipH continuing
        if sc == 6:
            # A.6-9/-8: no-op object op leaves `selected`; reset_sel keeps stale object planes
            cyc = s % 5
            if cyc == 0:
                m[1:3, 1:4] = 1
                return 24, m                       # Rotate90 fresh selection (odd/even dims -> parity)
            if cyc == 1:
                return 25, m                       # Rotate270 continuing
            if cyc == 2:
                return 5, m                        # Color5 with empty selection: only reset_sel happens
            if cyc == 3:
                return 22, m                       # MoveR while inactive and nothing selected: total no-op
            m[2, 2] = 1
            return 10 + (s % 10), m                # FloodFill from a point
        # sc == 7: A.6-10 out-of-contract mask values
        cyc = s % 4
        if cyc == 0:
            m[1, 1] = 2                            # single 2: FloodFill no-op (sum != 1), Color paints
            return 12 if s % 8 == 0 else 4, m
        if cyc == 1:
            m[1, 1] = 2
            m[1, 2] = -1                           # sum == 1, argmax -> (1,1)
            return 13, m
        if cyc == 2:
            m[0:2, 0:2] = 1
            m[1, 1] = -1                           # truthy but not >0: in bbox, not in object
            return 23, m
        m[2, 1] = -1
        return 28, m                               # CopyI: any(sel>0) false -> no-op


# ---- large same-colour regions for FloodFill (the reference's recursive DFS, color.py:8-30, is the pin) -------------------
class FloodScript:
    """Tasks: stripes, coarse 3-colour blobs, a 1-wide spiral corridor (long, winding regions); actions: FloodFill from a random
    in-bounds point most of the time, now and then a Color rectangle or a CopyFromInput so that regions change and come back."""

    def task(self, n, H, W):
        rng = RD.SplitMix64(900 + n)
        kind = n % 3
        ii, jj = np.arange(H)[:, None], np.arange(W)[None, :]
        if kind == 0:
            p = 2 + rng.below(4)
            g = (((ii // p) + (jj // p if rng.below(3) == 0 else 0)) % 2) * (1 + rng.below(9))
        elif kind == 1:
            coarse = np.array([[rng.below(3) for _ in range(W // 5 + 1)] for _ in range(H // 5 + 1)])
            g = np.kron(coarse, np.ones((5, 5), np.int64))[:H, :W] + 1
        else:
            g = np.full((H, W), 2)
            top, left, bot, right = 0, 0, H - 1, W - 1
            while top <= bot and left <= right:
                g[top, left:right + 1] = 1
                g[top:bot + 1, right] = 1
                if bot > top + 1:
                    g[bot, left + 2:right + 1] = 1
                if right > left + 2 and bot > top + 2:
                    g[top + 2:bot + 1, left + 2] = 1
                top, left, bot, right = top + 2, left + 2, bot - 2, right - 2
                if top <= bot and left <= right:
                    g[top, left] = 1
        a = np.asarray(g, np.int8)
        return a, a.copy()

    def __call__(self, rng, s, n, H, W, n_ops):
        m = np.zeros((H, W), np.int8)
        r = rng.below(10)
        if r < 7:
            m[rng.below(H), rng.below(W)] = 1
            return 10 + rng.below(10), m            # FloodFill c from a point
        if r == 7:
            x, y = rng.below(H), rng.below(W)
            m[x:x + 1 + rng.below(6), y:y + 1 + rng.below(6)] = 1
            return rng.below(10), m                 # Color c on a small rectangle (cuts corridors / merges regions)
        if r == 8:
            return 31, m                            # CopyFromInput: the original regions come back
        m[rng.below(H), rng.below(W)] = 1
        m[rng.below(H), rng.below(W)] = 1           # two seeds (or one, twice): FloodFill only acts when sum == 1
        return 10 + rng.below(10), m


def main():
    RD.import_reference()
    sys.setrecursionlimit(5000)  # (the reference's DFS recurses once per cell of the region)
    capture("o2arc_05", "o2arc", 5, 5, -1, 32, 128, 101)
    capture("o2arc_10", "o2arc", 10, 10, 3, 32, 128, 102)
    capture("o2arc_30", "o2arc", 30, 30, -1, 32, 160, 103)
    capture("o2arc_30_bool", "o2arc", 30, 30, 3, 16, 96, 112, bool_masks=True)
    capture("o2arc_20", "o2arc", 20, 20, -1, 16, 96, 113)
    capture("o2arc_30_t127", "o2arc", 30, 30, 127, 8, 96, 104, weird=True)
    capture("o2arc_crop_10", "o2arc_crop", 10, 10, -1, 8, 96, 105)
    capture("o2arc_exotic_12", "o2arc_exotic", 12, 12, -1, 8, 128, 106)
    capture("arc_10", "arc", 10, 10, 3, 8, 96, 107)
    capture("arc_30", "arc", 30, 30, 3, 8, 96, 108)
    capture("raw_05", "raw", 5, 5, -1, 8, 96, 109)
    capture("raw_30", "raw", 30, 30, 2, 4, 64, 110)
    capture("quirks_30", "o2arc", 30, 30, -1, 8, 300, 111, script=QuirkScript(), full_every=50)
    capture("flood_30", "o2arc", 30, 30, -1, 12, 48, 114, script=FloodScript(), full_every=12)
    capture("flood_17x21", "o2arc", 17, 21, -1, 9, 48, 115, script=FloodScript(), full_every=12)


if __name__ == "__main__":
    main()
