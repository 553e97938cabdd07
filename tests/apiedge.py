"""API-edge script runner: drives an ARCLE-style single env (the reference's classes when the fixture is generated,
arcle_amd's when it is replayed) through its PUBLIC Python API only and records what every call returns.

Test infrastructure.  The same code runs on both sides, so the fixture pins the Python-level semantics of the boundary
(/root/reference/arcle/envs/o2arcenv.py:121-151, arcenv.py:51-76,139-172, base.py:69-118,172-183), not only the cell arithmetic:

  * `operation` given as a negative list index (-1 ... -len), as np.int64 / np.int8 / float / 0-d and 1-element arrays;
    out-of-range indices (IndexError, state untouched)
  * selections of dtype bool / uint8 / int16 / int64 / float32 / float64 and non-contiguous layouts (Fortran order,
    strided slices of a larger array, reversed-then-restored views)
  * reset(options=...) with prob_index / subprob_index / adaptation / reset_on_submit, and option-less resets whose task
    draw comes from the global np.random stream (base.py:99,104; loader.py:50)
  * transition() and submit() on a deepcopy of the state (README.md:55) and on env.current_state itself
  * what the env keeps between calls: last_action_op, submit_count, action_steps, last_reward, info, and whether the
    observation returned IS env.current_state / the previous observation (one dict per episode in the reference)

Script = JSON-able dict; record = dict of arrays.  See tests/golden/make_golden_api.py for the generator.
"""
import copy
import json

import numpy as np

NONE = -999  # (marker for a None scalar in the int64 record columns)
COLS = ("reward", "terminated", "truncated", "info_steps", "info_submit_count", "last_action_op", "submit_count", "action_steps",
        "last_reward", "obs_is_current", "obs_is_prev", "exc")
EXC = {None: 0, "IndexError": 1, "ValueError": 2, "AssertionError": 3, "TypeError": 4, "KeyError": 5}


def flatten(state):
    """State dict -> int64 vector, keys sorted at both levels (values, not dtypes: a bool `selected` compares as 0/1)."""
    parts = []
    for k in sorted(state):
        v = state[k]
        if isinstance(v, dict):
            for k2 in sorted(v):
                parts.append(np.asarray(v[k2]).astype(np.int64).ravel())
        else:
            parts.append(np.asarray(v).astype(np.int64).ravel())
    return np.concatenate(parts)


def make_operation(value, how):
    """The `operation` entry of an action in one of the forms callers use."""
    if how == "int":
        return int(value)
    if how == "int64":
        return np.int64(value)
    if how == "int8":
        return np.int8(value)
    if how == "int32":
        return np.int32(value)
    if how == "float":
        return float(value)
    if how == "arr0":
        return np.array(value)
    if how == "arr1":
        return np.array([value])  # int() of a 1-element array (deprecated in NumPy, still accepted)
    raise KeyError(how)


def make_selection(values, dtype, layout):
    """`values`: [H,W] integer array; returns the same values as dtype `dtype` in memory layout `layout`."""
    a = np.asarray(values).astype(np.dtype(dtype))
    H, W = a.shape
    if layout == "c":
        return np.ascontiguousarray(a)
    if layout == "f":
        return np.asfortranarray(a)
    if layout == "strided":       # every second element of a larger buffer, in both axes
        big = np.zeros((2 * H + 1, 2 * W + 3), a.dtype)
        big[1::2, 2::2][:H, :W] = a
        return big[1::2, 2::2][:H, :W]
    if layout == "reversed":      # negative strides
        return np.ascontiguousarray(a[::-1, ::-1])[::-1, ::-1]
    if layout == "transposed":    # a transposed view of the transposed data
        return np.ascontiguousarray(a.T).T
    if layout == "readonly":
        b = np.ascontiguousarray(a)
        b.setflags(write=False)
        return b
    raise KeyError(layout)


def _action(call, sels):
    return {"selection": make_selection(sels[call["sel"]], call.get("dtype", "int8"), call.get("layout", "c")),
            "operation": make_operation(call["op"], call.get("op_as", "int"))}


def _scalars(env, out, obs, prev_obs, exc=None):
    info = getattr(env, "info", None) or {}
    g = lambda v: NONE if v is None else int(v)  # noqa: E731
    return [g(out[1]) if out else NONE, g(out[2]) if out else NONE, g(out[3]) if out else NONE,
            g(info.get("steps")), g(info.get("submit_count")), g(getattr(env, "last_action_op", None)),
            g(getattr(env, "submit_count", None)), g(getattr(env, "action_steps", None)), g(getattr(env, "last_reward", None)),
            NONE if obs is None else int(obs is env.current_state), NONE if obs is None or prev_obs is None else int(obs is prev_obs),
            EXC[exc]]


def run_scenario(env, sc, sels):
    """Runs one scenario on `env`.  Returns (rows int64 [calls, L], aux int64 [calls, L] — the deepcopy a transition ran on, or the
    state again —, scal int64 [calls, len(COLS)], info arrays of the resets)."""
    rows, aux, scal, infos = [], [], [], []
    prev_obs = None
    for call in sc["calls"]:
        k = call["k"]
        exc = None
        out = None
        obs = None
        side = None
        try:
            if k == "reset":
                if call.get("np_seed") is not None:
                    np.random.seed(call["np_seed"])
                obs, info = env.reset(options=copy.deepcopy(call.get("options")))
                infos.append(np.concatenate([np.asarray(info["input"]).astype(np.int64).ravel(), np.asarray(info["input_dim"], np.int64),
                                             np.asarray(info["answer"]).astype(np.int64).ravel(), np.asarray(info["answer_dim"], np.int64)]))
            elif k == "step":
                out = env.step(_action(call, sels))
                obs = out[0]
            elif k in ("transition", "submit"):
                fn = env.transition if k == "transition" else env.submit
                if call["on"] == "deepcopy":
                    side = copy.deepcopy(env.current_state)
                    fn(side, _action(call, sels))
                else:
                    fn(env.current_state, _action(call, sels))
            else:
                raise KeyError(k)
        except (IndexError, ValueError, AssertionError, TypeError) as e:
            exc = type(e).__name__
        cur = flatten(env.current_state)
        rows.append(cur)
        aux.append(flatten(side) if side is not None else cur)
        scal.append(_scalars(env, out, obs, prev_obs, exc))
        if obs is not None:
            prev_obs = obs
    return np.stack(rows), np.stack(aux), np.array(scal, np.int64), (np.stack(infos) if infos else np.zeros((0, 1), np.int64))


class _Tasks:
    """Loader payload: a list of (ex_in, ex_out, tt_in, tt_out, desc) built from the fixture's padded arrays."""

    def __init__(self, tasks):
        self.tasks = tasks

    def as_parse_result(self):
        return [([np.array(a, np.int8) for a in t["ex_in"]], [np.array(a, np.int8) for a in t["ex_out"]],
                 [np.array(a, np.int8) for a in t["tt_in"]], [np.array(a, np.int8) for a in t["tt_out"]], {"id": t["id"]})
                for t in self.tasks]


def make_loader(LoaderBase, tasks):
    payload = _Tasks(tasks).as_parse_result()

    class FixtureLoader(LoaderBase):
        def get_path(self, **kw):
            return [""]

        def parse(self, **kw):
            return payload
    return FixtureLoader()


def run_script(script, make_env, sel_pools):
    """make_env(cls_name, loader_tasks, H, W, max_trial) -> env; sel_pools[i]: [n,H,W] integer masks of scenario i (kept in the .npz
    as `s{i}_sels`, the calls refer to them by index).  Returns {name: array} for np.savez."""
    out = {}
    for i, sc in enumerate(script["scenarios"]):
        sels = np.asarray(sel_pools[i]).astype(np.int64)
        env = make_env(sc["cls"], script["tasks"], sc["H"], sc["W"], sc["max_trial"])
        rows, aux, scal, infos = run_scenario(env, sc, sels)
        out[f"s{i}_rows"], out[f"s{i}_aux"], out[f"s{i}_scal"], out[f"s{i}_info"] = rows, aux, scal, infos
    return out


def compare(script, got, want):
    """List of human-readable mismatches between two run_script results."""
    errs = []
    for i, sc in enumerate(script["scenarios"]):
        tag = f"s{i}[{sc['name']}]"
        for part in ("rows", "aux", "info"):
            a, b = got[f"s{i}_{part}"], want[f"s{i}_{part}"]
            if a.shape != b.shape:
                errs.append(f"{tag} {part}: shape {a.shape} != {b.shape}")
                continue
            bad = np.flatnonzero((a != b).any(axis=1)) if a.size else []
            for c in bad[:3]:
                what = json.dumps(sc["calls"][c]) if part != "info" else "reset"
                errs.append(f"{tag} {part} call {c} {what}: {int((a[c] != b[c]).sum())} cells differ (first at {int(np.flatnonzero(a[c] != b[c])[0])})")
        a, b = got[f"s{i}_scal"], want[f"s{i}_scal"]
        for c, j in zip(*np.nonzero(a != b)):
            errs.append(f"{tag} call {c} {json.dumps(sc['calls'][c])}: {COLS[j]} = {a[c, j]}, reference {b[c, j]}")
    return errs
