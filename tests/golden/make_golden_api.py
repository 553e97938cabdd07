#!/usr/bin/env python3
"""API-edge golden vectors: the reference's OWN env classes (unmodified /root/reference, imported through oracle/stubs) driven
through the Python-level corners of the single-env boundary by tests/apiedge.py.  Build-container only; the outputs
tests/golden/api_edge.npz (arrays) and tests/golden/api_edge_script.json (the calls, pure data) are what travels.

    python tests/golden/make_golden_api.py

What is pinned (reference file:line):
  neg_*       self.operations[op] with op = -1 ... -len (o2arcenv.py:149-151, arcenv.py:62-68): the op runs, `last_action_op`
              keeps the NEGATIVE index, so reward() (o2arcenv.py:121-128) is 0 even for Submit-by-(-1) on a solved grid
  op_types    int(action['operation']) for np.int64 / np.int8 / np.int32 / float / 0-d / 1-element arrays; 35 and -36 -> IndexError
  sel_dtypes  bool / uint8 / int16 / int32 / int64 / float32 / float64 masks, C / Fortran / strided / negative-stride / transposed /
              read-only layouts; signed and float masks also carry the out-of-contract values {2, 3, -1} (make_golden.py:206)
  reset_*     reset(options={prob_index, subprob_index, adaptation, reset_on_submit}) (base.py:87-118) on a 3-task loader with
              distinct demo / test pairs; option-less resets drawing the task from np.random (loader.py:50, base.py:99,104);
              bad indices (AssertionError / IndexError)
  transition  transition() / submit() on a deepcopy of the state (README.md:55) and on env.current_state; Submit through
              transition() counts in env.submit_count (base.py:175) but not in action_steps
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import refdriver as RD  # noqa: E402
import apiedge as AE  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def _grid(rng, h, w, style):
    if style == 0:
        return [[int(rng.below(10)) for _ in range(w)] for _ in range(h)]
    c0, c1 = int(rng.below(10)), int(rng.below(10))
    return [[c0 if rng.chance(2, 3) else c1 for _ in range(w)] for _ in range(h)]


def build_tasks(rng):
    """3 tasks; task 0: one demo pair with answer == input (solved from reset) and one test pair with a different answer; tasks 1, 2:
    2-3 demo pairs and 1-2 test pairs of assorted sizes (all sides <= 7 so they fit the small scenarios too)."""
    tasks = []
    a = _grid(rng, 5, 6, 1)
    b = _grid(rng, 4, 4, 0)
    tasks.append({"id": "t0", "ex_in": [a], "ex_out": [a], "tt_in": [b], "tt_out": [_grid(rng, 4, 4, 0)]})
    for t in (1, 2):
        n_ex, n_tt = 2 + rng.below(2), 1 + rng.below(2)
        mk = lambda: _grid(rng, 1 + rng.below(7), 1 + rng.below(7), rng.below(2))  # noqa: E731
        ex_in = [mk() for _ in range(n_ex)]
        ex_out = [g if rng.chance(1, 2) else mk() for g in ex_in]
        tt_in = [mk() for _ in range(n_tt)]
        tt_out = [g if rng.chance(1, 2) else mk() for g in tt_in]
        tasks.append({"id": f"t{t}", "ex_in": ex_in, "ex_out": ex_out, "tt_in": tt_in, "tt_out": tt_out})
    return tasks


class SelPool:
    def __init__(self, rng, H, W):
        self.rng, self.H, self.W, self.sels = rng, H, W, []

    def add(self, m):
        self.sels.append(np.asarray(m, np.int64).tolist())
        return len(self.sels) - 1

    def random(self, exotic=False, small=True):
        rng, H, W = self.rng, self.H, self.W
        m = np.zeros((H, W), np.int64)
        t = rng.below(10)
        lim_h, lim_w = (min(H, 7), min(W, 7)) if small else (H, W)
        if t < 4:      # rectangle
            x1, y1 = rng.below(lim_h), rng.below(lim_w)
            x2, y2 = min(H - 1, x1 + rng.below(4)), min(W - 1, y1 + rng.below(4))
            m[x1:x2 + 1, y1:y2 + 1] = 1
        elif t < 6:    # point
            m[rng.below(lim_h), rng.below(lim_w)] = 1
        elif t < 7:    # empty
            pass
        else:          # scatter
            for _ in range(1 + rng.below(12)):
                m[rng.below(lim_h), rng.below(lim_w)] = 1
        if exotic and rng.chance(1, 2):
            for _ in range(1 + rng.below(3)):
                m[rng.below(lim_h), rng.below(lim_w)] = [2, -1, 3, 1][rng.below(4)]
        return self.add(m)


def o2arc_pick(rng, n):
    return RD.pick_op(rng, n, "o2arc")


def scenario_neg(rng, cls, H, W, n_ops, max_trial):
    pool = SelPool(rng, H, W)
    calls = [{"k": "reset", "options": {"prob_index": 0, "subprob_index": 0}}]
    # Submit by index -1 on the solved grid, then by its positive index after a reset
    calls.append({"k": "step", "op": -1, "sel": pool.random()})
    calls.append({"k": "step", "op": -1, "sel": pool.random()})
    calls.append({"k": "reset", "options": {"prob_index": 0, "subprob_index": 0}})
    calls.append({"k": "step", "op": n_ops - 1, "sel": pool.random()})
    calls.append({"k": "reset", "options": {"prob_index": 0, "subprob_index": 0}})
    # every negative index once, interleaved with positive ones so the ops have something to do
    order = list(range(-n_ops, 0))
    for j in range(len(order) - 1, 0, -1):  # Fisher-Yates on the splitmix stream
        i = rng.below(j + 1)
        order[i], order[j] = order[j], order[i]
    for op in order:
        if op == -1:
            continue
        if rng.chance(1, 2):
            calls.append({"k": "step", "op": int(rng.below(n_ops - 1)), "sel": pool.random()})
        calls.append({"k": "step", "op": op, "sel": pool.random(), "op_as": ["int", "int64", "int8"][rng.below(3)]})
    # unsolved grid, Submit by -1, then solved again through CopyFromInput-equivalent and Submit by -1 and by its index
    calls.append({"k": "step", "op": -1, "sel": pool.random()})
    calls.append({"k": "reset", "options": {"prob_index": 0, "subprob_index": 0}})
    calls.append({"k": "step", "op": 3, "sel": pool.random()})
    calls.append({"k": "step", "op": -1, "sel": pool.random()})
    calls.append({"k": "step", "op": n_ops - 1, "sel": pool.random()})
    # out of range either way
    calls.append({"k": "step", "op": n_ops, "sel": pool.random()})
    calls.append({"k": "step", "op": -n_ops - 1, "sel": pool.random()})
    calls.append({"k": "step", "op": 0, "sel": pool.random()})
    return {"name": f"neg_{cls}_{H}x{W}", "cls": cls, "H": H, "W": W, "max_trial": max_trial, "sels": pool.sels, "calls": calls}


def scenario_op_types(rng, H, W):
    pool = SelPool(rng, H, W)
    calls = [{"k": "reset", "options": {"prob_index": 1, "subprob_index": 0}}]
    for how in ("int", "int64", "int8", "int32", "float", "arr0", "arr1"):
        for _ in range(5):
            op = o2arc_pick(rng, 35)
            if rng.chance(1, 4):
                op -= 35
            calls.append({"k": "step", "op": int(op), "op_as": how, "sel": pool.random()})
    calls.append({"k": "step", "op": 34, "op_as": "float", "sel": pool.random()})
    return {"name": f"op_types_{H}x{W}", "cls": "o2arc", "H": H, "W": W, "max_trial": 3, "sels": pool.sels, "calls": calls}


def scenario_sel_dtypes(rng, cls, H, W, n_ops):
    pool = SelPool(rng, H, W)
    calls = [{"k": "reset", "options": {"prob_index": 2, "subprob_index": 0}}]
    dtypes = ["bool", "uint8", "int8", "int16", "int32", "int64", "float32", "float64"]
    layouts = ["c", "f", "strided", "reversed", "transposed", "readonly"]
    for d in dtypes:
        for lay in layouts:
            for _ in range(3):
                exotic = d in ("int8", "int16", "int32", "int64", "float32", "float64")
                op = o2arc_pick(rng, n_ops) if cls == "o2arc" else int(rng.below(n_ops))
                if op == n_ops - 1 and rng.chance(3, 4):
                    op = int(rng.below(n_ops - 1))
                calls.append({"k": "step", "op": int(op), "sel": pool.random(exotic=exotic), "dtype": d, "layout": lay})
        if cls != "raw":
            calls.append({"k": "transition", "on": "deepcopy", "op": int(rng.below(n_ops - 1)), "sel": pool.random(), "dtype": d, "layout": "f"})
    return {"name": f"sel_dtypes_{cls}_{H}x{W}", "cls": cls, "H": H, "W": W, "max_trial": -1, "sels": pool.sels, "calls": calls}


def scenario_reset(rng, cls, H, W, n_ops):
    pool = SelPool(rng, H, W)
    calls = []
    opts = [
        {"prob_index": 0}, {"prob_index": 1, "subprob_index": 1}, {"prob_index": 2, "subprob_index": 0, "adaptation": False},
        {"prob_index": 1, "adaptation": False}, {"prob_index": 1, "adaptation": 0}, {"prob_index": 2, "adaptation": None},
        {"prob_index": 0, "subprob_index": 0, "reset_on_submit": True}, {"prob_index": 1, "subprob_index": -1},
        {"prob_index": 0, "subprob_index": 0, "adaptation": False, "reset_on_submit": True},
        None, {}, None, {"adaptation": False}, {"subprob_index": 0}, None,
    ]
    seed = 1000
    for o in opts:
        seed += 17
        calls.append({"k": "reset", "options": o, "np_seed": seed})
        for _ in range(4):
            op = o2arc_pick(rng, n_ops) if cls == "o2arc" else int(rng.below(n_ops))
            calls.append({"k": "step", "op": int(op), "sel": pool.random()})
        calls.append({"k": "step", "op": n_ops - 1, "sel": pool.random()})   # Submit (resets the state under reset_on_submit)
        calls.append({"k": "step", "op": int(rng.below(10)), "sel": pool.random()})
        calls.append({"k": "step", "op": n_ops - 1, "sel": pool.random()})
    # bad indices: the reference asserts on the task index (loader.py:54) and indexes a list with the pair index
    calls.append({"k": "reset", "options": {"prob_index": 3}, "np_seed": 5})
    calls.append({"k": "reset", "options": {"prob_index": -1}, "np_seed": 5})
    calls.append({"k": "reset", "options": {"prob_index": 0, "subprob_index": 4}, "np_seed": 5})
    calls.append({"k": "reset", "options": {"prob_index": 2, "subprob_index": 0}, "np_seed": 5})
    calls.append({"k": "step", "op": 1, "sel": pool.random()})
    return {"name": f"reset_{cls}_{H}x{W}", "cls": cls, "H": H, "W": W, "max_trial": 2, "sels": pool.sels, "calls": calls}


def scenario_transition(rng, cls, H, W, n_ops, max_trial, ros):
    pool = SelPool(rng, H, W)
    calls = [{"k": "reset", "options": {"prob_index": 0, "subprob_index": 0, "reset_on_submit": ros}}]
    for r in range(60):
        op = o2arc_pick(rng, n_ops) if cls == "o2arc" else int(rng.below(n_ops))
        t = rng.below(10)
        if cls == "raw" and t < 9:  # (the reference's RawARCEnv has no transition(): arcenv.py:17-76)
            t = 0 if t < 6 else 9
        if t < 4:
            calls.append({"k": "step", "op": int(op), "sel": pool.random()})
        elif t < 7:
            calls.append({"k": "transition", "on": "deepcopy", "op": int(op - n_ops if rng.chance(1, 5) else op), "sel": pool.random()})
        elif t < 9:
            calls.append({"k": "transition", "on": "live", "op": int(op), "sel": pool.random()})
        else:
            calls.append({"k": "submit", "on": "deepcopy" if rng.chance(1, 2) else "live", "op": int(rng.below(n_ops)), "sel": pool.random()})
        if r == 30:
            calls.append({"k": "reset", "options": {"prob_index": 1, "subprob_index": 0, "reset_on_submit": ros}})
    if cls != "raw":
        calls.append({"k": "transition", "on": "deepcopy", "op": n_ops, "sel": pool.random()})  # IndexError
    calls.append({"k": "step", "op": 2, "sel": pool.random()})
    return {"name": f"transition_{cls}_{H}x{W}_ros{int(ros)}", "cls": cls, "H": H, "W": W, "max_trial": max_trial, "sels": pool.sels,
            "calls": calls}


def build_script():
    rng = RD.SplitMix64(20260929)
    tasks = build_tasks(rng)
    sc = [
        scenario_neg(rng, "o2arc", 30, 30, 35, -1), scenario_neg(rng, "o2arc", 8, 9, 35, 3),
        scenario_neg(rng, "arc", 30, 30, 27, 3), scenario_neg(rng, "raw", 30, 30, 12, -1), scenario_neg(rng, "raw", 7, 7, 12, 2),
        scenario_op_types(rng, 30, 30),
        scenario_sel_dtypes(rng, "o2arc", 30, 30, 35), scenario_sel_dtypes(rng, "o2arc", 9, 8, 35),
        scenario_sel_dtypes(rng, "arc", 12, 12, 27), scenario_sel_dtypes(rng, "raw", 30, 30, 12),
        scenario_reset(rng, "o2arc", 30, 30, 35), scenario_reset(rng, "arc", 10, 10, 27), scenario_reset(rng, "raw", 8, 8, 12),
        scenario_transition(rng, "o2arc", 30, 30, 35, -1, False), scenario_transition(rng, "o2arc", 10, 10, 35, 5, False),
        scenario_transition(rng, "o2arc", 30, 30, 35, 4, True),
        scenario_transition(rng, "arc", 30, 30, 27, 3, False), scenario_transition(rng, "raw", 30, 30, 12, -1, False),
        scenario_transition(rng, "raw", 9, 9, 12, 3, True),
    ]
    return {"tasks": tasks, "scenarios": sc}


def make_reference_env(cls, tasks, H, W, max_trial):
    RD.import_reference()
    from arcle.loaders import Loader
    from arcle.envs import O2ARCv2Env, RawARCEnv, ARCEnv

    class ARCEnv27(ARCEnv):  # (arcenv.py:120 leaves 8 None slots -> base.py:66 AttributeError; the 27 installed ops)
        def create_operations(self):
            return super().create_operations()[:27]

    klass = {"o2arc": O2ARCv2Env, "arc": ARCEnv27, "raw": RawARCEnv}[cls]
    return klass(data_loader=AE.make_loader(Loader, tasks), max_grid_size=(H, W), colors=10, max_trial=max_trial)


def main():
    import warnings
    warnings.simplefilter("ignore")
    script = build_script()
    pools = [np.array(s.pop("sels"), np.int8) for s in script["scenarios"]]
    rec = AE.run_script(script, make_reference_env, pools)
    for i, p in enumerate(pools):
        rec[f"s{i}_sels"] = p
    with open(os.path.join(OUT, "api_edge_script.json"), "w") as f:
        json.dump(script, f, separators=(",", ":"))
    np.savez_compressed(os.path.join(OUT, "api_edge.npz"), **rec)
    n_calls = sum(len(s["calls"]) for s in script["scenarios"])
    exc = sum(int((rec[f"s{i}_scal"][:, -1] != 0).sum()) for i in range(len(script["scenarios"])))
    print(f"api_edge: {len(script['scenarios'])} scenarios, {n_calls} calls, {exc} raising; "
          f"{os.path.getsize(os.path.join(OUT, 'api_edge.npz'))} + {os.path.getsize(os.path.join(OUT, 'api_edge_script.json'))} bytes")
    for i, s in enumerate(script["scenarios"]):
        sc = rec[f"s{i}_scal"]
        print(f"  {s['name']:34s} calls {len(s['calls']):4d}  exc {int((sc[:, -1] != 0).sum()):2d}  rewards {int((sc[:, 0] == 1).sum()):2d}  "
              f"terminated {int((sc[:, 1] == 1).sum()):3d}")


if __name__ == "__main__":
    main()
