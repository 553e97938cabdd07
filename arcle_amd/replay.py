"""O2ARC human-trace replayer — the validation harness of the reference (tests/o2arc_check.py) on the device path.

An O2ARC web-UI log is a list of entries `(timestamp, action_name, data, grid_after)`; the reference converts every entry
to `(operation, selection)` (`action_convert`, o2arc_check.py:21-99), replaces the selection of an object operation by
an empty one when it equals the env's current `selected` plane (the object is continued, :169-170), steps the env and
compares `grid[:h,:w]` with the logged grid (:184-195).  Here many traces are replayed at once, one env per trace: the
selections go to the device as logged and the continuation rule is applied by the step kernel
(ARCLE_STEP_CONTINUE_RULE).  The human-trace pickles and the ARC datasets are not part of the reference checkout
(SURVEY.md §4), so the pin is a set of synthetic logs replayed through the reference's own harness logic
(tests/golden/make_golden_research.py).
"""
import numpy as np
import torch

from .engine import STEP_CONTINUE_RULE
from .envs import ARCVecEnv, O2ARCv2Env
from .loaders import Loader

MOVE = {"U": 20, "D": 21, "R": 22, "L": 23}
BOXED = {"FlipX": 27, "FlipY": 26, "RotateCW": 25, "RotateCCW": 24}  # op numbering of o2arcenv.py:88-113
COPY = {"Input Grid": 28, "Output Grid": 29}


def action_convert(entry, H=30, W=30):
    """One O2ARC log entry -> (operation index of the 35-op O2ARCv2Env table, bool selection mask [H, W])."""
    _, name, data, _ = entry
    sel = np.zeros((H, W), np.bool_)

    def box():
        (h0, w0), (h1, w1) = data[0], data[1]
        sel[h0:h1 + 1, w0:w1 + 1] = True

    if name == "CopyFromInput":
        return 31, sel
    if name == "ResetGrid":
        return 32, sel
    if name == "Submit":
        return 34, sel
    if name == "ResizeGrid":
        h, w = data[0]
        sel[:h, :w] = True
        return 33, sel
    if name == "Color":  # one pixel, colour data[1]
        sel[data[0][0], data[0][1]] = True
        return int(data[1]), sel
    if name == "Fill":  # a rectangle painted with colour data[2]
        box()
        return int(data[2]), sel
    if name in BOXED:
        box()
        return BOXED[name], sel
    if name == "Move":
        box()
        return MOVE[data[2]], sel
    if name == "Copy":
        box()
        return COPY[data[2]], sel
    if name == "Paste":
        sel[data[0][0], data[0][1]] = True
        return 30, sel
    if name == "FloodFill":
        sel[data[0][0], data[0][1]] = True
        return 10 + int(data[1]), sel
    raise ValueError(f"unknown O2ARC action {name!r}")


class _PairLoader(Loader):
    """One problem per trace: (input, answer) of the pair the trace was recorded on."""

    def __init__(self, pairs):
        self._pairs = pairs
        super().__init__()

    def get_path(self, **kwargs):
        return [""]

    def parse(self, **kwargs):
        return [([i], [o], [i], [o], {"id": f"trace{n}"}) for n, (i, o) in enumerate(self._pairs)]


def replay_traces(traces, pairs, device=None, expected=None):
    """traces: list of O2ARC logs (lists of entries); pairs: list of (input, answer) arrays, one per trace.
    Steps every trace on its own env (30x30 O2ARCv2Env, test pairs, adaptation=False as o2arc_check.py:148) and returns a
    list per trace of (grid[:h,:w]) after every step; with `expected` (a list per trace of logged grids) also the index
    of the first mismatching step per trace (-1 = the whole trace reproduces)."""
    n = len(traces)
    T = max(len(t) for t in traces)
    venv = ARCVecEnv(O2ARCv2Env, n, _PairLoader(pairs), max_grid_size=(30, 30), device=device)
    venv.reset(options={"adaptation": False, "prob_index": np.arange(n), "subprob_index": 0})
    ops = np.full((T, n), 32, np.int32)  # a finished trace idles on ResetGrid: nobody looks at it any more
    sels = np.zeros((T, n, 30, 30), np.int8)
    for i, tr in enumerate(traces):
        for t, entry in enumerate(tr):
            ops[t, i], m = action_convert(entry)
            sels[t, i] = m
    flags = venv.flags | STEP_CONTINUE_RULE
    grids = [[] for _ in range(n)]
    first_bad = [-1] * n
    dev_ops, dev_sels = torch.from_numpy(ops).to(venv.device), torch.from_numpy(sels).to(venv.device)
    for t in range(T):
        venv.batch.step_mask(dev_sels[t], dev_ops[t], flags)
        g = venv.batch.plane("grid").cpu().numpy()
        d = venv.batch.field("grid_dim").cpu().numpy()
        for i, tr in enumerate(traces):
            if t >= len(tr):
                continue
            cur = g[i, :d[i, 0], :d[i, 1]].copy()
            grids[i].append(cur)
            if expected is not None and first_bad[i] < 0:
                want = np.asarray(expected[i][t])
                if want.shape != cur.shape or np.any(want.astype(np.int8) != cur):  # o2arc_check.py:185
                    first_bad[i] = t
    venv.check_errors()
    return (grids, first_bad) if expected is not None else grids
