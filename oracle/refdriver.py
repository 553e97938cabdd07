"""Drives the UNMODIFIED reference (imported from /root/reference with the stubs in
oracle/stubs) — build-container only, test infrastructure only.

Used by oracle/diff_vs_reference.py (differential fuzz of oracle/arcle_oracle.c against the
reference) and tests/golden/make_golden.py (captures the committed golden vectors).  Nothing
here is importable on the GPU box (no /root/reference there) and nothing in arcle_amd/, bench.py
or the gpu tests imports it.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("ARCLE_REFERENCE_ROOT", "/root/reference")


def import_reference():
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "arcle")):
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    for p in (os.path.join(_HERE, "stubs"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import arcle  # noqa: F401
    return arcle


# ---- deterministic stream (splitmix64), independent of numpy's generators ----------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n

    def chance(self, num, den):
        return self.below(den) < num


def random_task(rng, H, W):
    """(input, answer): un-padded int8 arrays like Loader.parse yields (loader.py:95-108)."""
    ih, iw = 1 + rng.below(H), 1 + rng.below(W)
    style = rng.below(4)
    if style == 0:    # uniform colours
        a = np.array([[rng.below(10) for _ in range(iw)] for _ in range(ih)], np.int8)
    elif style == 1:  # half zeros
        a = np.array([[rng.below(10) if rng.chance(1, 2) else 0 for _ in range(iw)] for _ in range(ih)], np.int8)
    elif style == 2:  # two colours, large connected regions (flood-fill food)
        c0, c1 = rng.below(10), rng.below(10)
        a = np.array([[c0 if rng.chance(2, 3) else c1 for _ in range(iw)] for _ in range(ih)], np.int8)
    else:             # stripes
        c0, c1 = rng.below(10), rng.below(10)
        a = np.array([[c0 if (i // 2) % 2 == 0 or j == (0 if (i // 4) % 2 else iw - 1) else c1
                       for j in range(iw)] for i in range(ih)], np.int8)
    if rng.chance(1, 2):
        ans = a.copy()
        if rng.chance(1, 2) and ans.size:
            ans[rng.below(ih), rng.below(iw)] = rng.below(10)
    else:
        ah, aw = 1 + rng.below(H), 1 + rng.below(W)
        ans = np.array([[rng.below(10) for _ in range(aw)] for _ in range(ah)], np.int8)
    return a, ans


_WRAPPERS = {}


def reference_wrapper_mask(H, W, kind, payload):
    """The selection mask the REFERENCE's BBoxWrapper / PointWrapper builds for a tuple (arcle/wrappers/bbox.py:22-30,
    43-49) — called on the reference classes themselves, wrapped around a reference env of that grid size."""
    if (H, W) not in _WRAPPERS:
        import_reference()
        from arcle.wrappers import BBoxWrapper, PointWrapper
        env = make_reference_env("o2arc", H, W, -1, (np.ones((1, 1), np.int8), np.ones((1, 1), np.int8)))
        _WRAPPERS[(H, W)] = (BBoxWrapper(env), PointWrapper(env))
    bw, pw = _WRAPPERS[(H, W)]
    act = (bw if kind == "bbox" else pw).action(tuple(int(v) for v in payload) + (0,))
    return np.asarray(act["selection"], np.int8).copy()


def random_selection(rng, H, W, weird=False):
    """Returns (kind, payload, mask): kind in {'bbox','point','mask'}; for a tuple the mask is what the reference's
    wrapper class builds from it (reference_wrapper_mask), otherwise the raw mask."""
    t = rng.below(100)
    m = np.zeros((H, W), np.int8)
    if t < 40:
        x1, y1, x2, y2 = rng.below(H), rng.below(W), rng.below(H), rng.below(W)
        if rng.chance(1, 2):  # small rectangles: keep objects on the grid more often
            x2 = min(H - 1, x1 + rng.below(4))
            y2 = min(W - 1, y1 + rng.below(4))
        return "bbox", (x1, y1, x2, y2), reference_wrapper_mask(H, W, "bbox", (x1, y1, x2, y2))
    if t < 60:
        x, y = rng.below(H), rng.below(W)
        return "point", (x, y), reference_wrapper_mask(H, W, "point", (x, y))
    if t < 80:
        return "mask", None, m  # empty selection
    if t < 97 or not weird:
        k = 1 + rng.below(6)
        for _ in range(k * (1 + rng.below(H))):
            m[rng.below(H), rng.below(W)] = 1
        return "mask", None, m
    # out-of-contract values (truthy but not 1; negative): SURVEY.md A.6-10
    for _ in range(1 + rng.below(3)):
        m[rng.below(H), rng.below(W)] = [2, -1, 1, 3][rng.below(4)]
    return "mask", None, m


# ---- op-table variants: (descriptor table for oracle/device, reference env factory) -------
def make_reference_env(variant, H, W, max_trial, task):
    """Builds a reference env of the given variant holding the single task `task`."""
    import_reference()
    from arcle.loaders import Loader
    from arcle.envs import O2ARCv2Env, RawARCEnv, ARCEnv
    from arcle.actions.object import reset_sel, keep_sel, gen_flip, gen_rotate, gen_paste, gen_copy, gen_move
    from arcle.actions.critical import crop_grid
    from arcle.actions.color import gen_color

    ti, to = task

    class OneTask(Loader):  # the reference's own fake-loader pattern, tests/o2arcex.py:10-21
        def get_path(self, **kw):
            return [""]

        def parse(self, **kw):
            return [([ti], [to], [ti], [to], {"id": "synthetic"})]

    class ARCEnv27(ARCEnv):  # arcenv.py:120 leaves 8 None slots -> base.py:66 AttributeError
        def create_operations(self):
            return super().create_operations()[:27]

    class O2ARCCrop(O2ARCv2Env):  # agents/env.py:23-28
        def create_operations(self):
            ops = super().create_operations()
            ops[33] = reset_sel(crop_grid)
            return ops

    class O2ARCExotic(O2ARCv2Env):
        """Exercises generators no shipped env installs: Rotate180, Flip D0/D1, keep_sel, paste_blank=False,
        un-wrapped colour, wrapped move."""
        def create_operations(self):
            ops = super().create_operations()
            ops[0] = gen_color(0)                      # no wrapper
            ops[1] = keep_sel(gen_color(1))
            ops[20] = reset_sel(gen_move(0))
            ops[24] = gen_rotate(2)
            ops[26] = gen_flip("D0")
            ops[27] = gen_flip("D1")
            ops[28] = keep_sel(gen_copy("I"))
            ops[30] = reset_sel(gen_paste(paste_blank=False))
            ops[33] = reset_sel(crop_grid)
            return ops

    cls = {"o2arc": O2ARCv2Env, "raw": RawARCEnv, "arc": ARCEnv27, "o2arc_crop": O2ARCCrop,
           "o2arc_exotic": O2ARCExotic}[variant]
    env = cls(data_loader=OneTask(), max_grid_size=(H, W), colors=10, max_trial=max_trial)
    env.reset(options={"prob_index": 0, "subprob_index": 0})
    return env


def variant_table(variant):
    """Descriptor table matching make_reference_env(variant) and the state kind it needs."""
    from oracle import oracle as O
    if variant in ("o2arc", "raw", "arc"):
        return variant, O.KIND_OPS[variant]()
    ops = O.o2arc_ops()
    R, K = O.F_RESET_SEL, O.F_KEEP_SEL
    if variant == "o2arc_crop":
        ops[33] = O.desc(O.OP_CROP_GRID, 0, R)
        return "o2arc", ops
    if variant == "o2arc_exotic":
        ops[0] = O.desc(O.OP_COLOR, 0)
        ops[1] = O.desc(O.OP_COLOR, 1, K)
        ops[20] = O.desc(O.OP_MOVE, 0, R)
        ops[24] = O.desc(O.OP_ROTATE, 2)
        ops[26] = O.desc(O.OP_FLIP, 2)
        ops[27] = O.desc(O.OP_FLIP, 3)
        ops[28] = O.desc(O.OP_COPY, 0, K)
        ops[30] = O.desc(O.OP_PASTE, 0, R)
        ops[33] = O.desc(O.OP_CROP_GRID, 0, R)
        return "o2arc", ops
    raise KeyError(variant)


def flatten_state(d):
    """Reference state dict -> flat {name: int8 array} with the oracle's field names."""
    out = {}
    for k, v in d.items():
        if k == "object_states":
            for k2, v2 in v.items():
                out[k2] = np.asarray(v2)
        else:
            out[k] = np.asarray(v)
    return out


def pick_op(rng, n_ops, variant):
    """Op mix that over-weights object ops, flood fill and clipboard ops."""
    if variant.startswith("o2arc"):
        t = rng.below(100)
        if t < 35:
            return 20 + rng.below(8)       # Move/Rotate/Flip
        if t < 50:
            return 10 + rng.below(10)      # FloodFill
        if t < 62:
            return 28 + rng.below(3)       # Copy/Paste
        if t < 70:
            return 31 + rng.below(3)       # critical
        if t < 74:
            return 34                      # Submit
        return rng.below(n_ops)
    return rng.below(n_ops)
