"""A plain-NumPy, one-env-at-a-time restatement of O2ARCv2Env.step() — the "CPU NumPy step()" baseline.

TEST INFRASTRUCTURE / BASELINE ONLY: used by tests/ (checked against oracle/arcle_oracle.c) and by bench.py's
`cpu_baseline.numpy_step` leg.  Nothing under arcle_amd/ imports it.

It is written from the behavioural spec in SURVEY.md Appendix A (not from the reference's files) with the same CALL
STRUCTURE as the reference's hot path (/root/reference/arcle/envs/o2arcenv.py:130-151): one Python `step()` per env
and per action, a table of Python closures indexed by the op, a state dict of small int8 ndarrays, and a handful of
small NumPy calls per op — which is where the reference spends its time (interpreter + NumPy-call
overhead).  FloodFill is an explicit-stack DFS in Python like the reference's recursive one (color.py:8-30).
"""
import numpy as np


def _bbox(m):
    rows = np.flatnonzero(m.any(axis=1))
    cols = np.flatnonzero(m.any(axis=0))
    return int(rows[0]), int(rows[-1]), int(cols[0]), int(cols[-1])


def _i8(v):
    return int(np.int8(np.uint8(v & 0xFF)))


class NumpyO2ARCEnv:
    """One O2ARCv2Env-like env (35-op table, A.3).  `step(action)` takes {'selection': HxW int8/bool, 'operation': int}."""

    def __init__(self, H=30, W=30, max_trial=-1):
        self.H, self.W, self.max_trial = H, W, max_trial
        self.operations = self._create_operations()

    # ---- reset (A.1) -----------------------------------------------------------------------------------
    def reset(self, task_input, task_answer):
        H, W = self.H, self.W
        z = lambda: np.zeros((H, W), np.int8)  # noqa: E731
        ih, iw = task_input.shape
        inp = z()
        inp[:ih, :iw] = task_input
        self.answer = np.asarray(task_answer, np.int8)
        self.state = {
            "input": inp, "input_dim": np.array([ih, iw], np.int8), "grid": inp.copy(),
            "grid_dim": np.array([ih, iw], np.int8), "selected": z(), "clip": z(), "clip_dim": np.zeros(2, np.int8),
            "trials_remain": np.array([self.max_trial], np.int8), "terminated": np.array([0], np.int8),
            "object_states": {"active": np.array([0], np.int8), "object": z(), "object_sel": z(), "background": z(),
                              "object_dim": np.zeros(2, np.int8), "object_pos": np.zeros(2, np.int8),
                              "rotation_parity": np.array([0], np.int8)},
        }
        self.action_steps = 0
        self.submit_count = 0
        return self.state

    # ---- helpers (A.2) -----------------------------------------------------------------------------------
    @staticmethod
    def _reset_sel(fn):
        def wrapped(state, action):
            state["selected"] = np.zeros_like(state["selected"])
            state["object_states"]["active"][0] = 0
            fn(state, action)
        return wrapped

    def _objsel(self, state, sel):
        o = state["object_states"]
        if np.any(sel):
            x0, x1, y0, y1 = _bbox(sel != 0)
            h, w = x1 - x0 + 1, y1 - y0 + 1
            part = sel[x0:x1 + 1, y0:y1 + 1] > 0
            o["object_dim"][:] = (h, w)
            o["object"] = np.zeros_like(state["grid"])
            o["object"][:h, :w] = np.where(part, state["grid"][x0:x1 + 1, y0:y1 + 1], 0)
            o["object_sel"] = np.zeros_like(state["grid"])
            o["object_sel"][:h, :w] = part
            o["background"] = np.where(sel > 0, 0, state["grid"]).astype(np.int8)
            o["object_pos"][:] = (x0, y0)
            o["active"][0] = 1
            o["rotation_parity"][0] = 0
            state["selected"] = np.asarray(sel, np.int8).copy()
            return True
        return bool(o["active"][0])

    @staticmethod
    def _place(state):
        o = state["object_states"]
        gh, gw = int(state["grid_dim"][0]), int(state["grid_dim"][1])
        x, y = int(o["object_pos"][0]), int(o["object_pos"][1])
        h, w = int(o["object_dim"][0]), int(o["object_dim"][1])
        state["grid"] = o["background"].copy()
        state["selected"] = np.zeros_like(state["grid"])
        if _i8(x + h) > 0 and x < gh and _i8(y + w) > 0 and y < gw:
            i0, i1, j0, j1 = max(0, x), min(gh, x + h), max(0, y), min(gw, y + w)
            patch = o["object"][i0 - x:i1 - x, j0 - y:j1 - y]
            np.copyto(state["grid"][i0:i1, j0:j1], patch, where=patch > 0)
            state["selected"][i0:i1, j0:j1] = o["object_sel"][i0 - x:i1 - x, j0 - y:j1 - y]

    # ---- ops (A.3, A.4) ------------------------------------------------------------------------------------
    def _create_operations(self):
        R = self._reset_sel
        ops = [R(self._gen_color(c)) for c in range(10)]
        ops += [R(self._gen_flood_fill(c)) for c in range(10)]
        ops += [self._gen_move(d) for d in range(4)]
        ops += [self._gen_rotate(1), self._gen_rotate(3), self._gen_flip(0), self._gen_flip(1)]
        ops += [R(self._gen_copy(False)), R(self._gen_copy(True)), R(self._paste)]
        ops += [R(self._copy_from_input), R(self._reset_grid), R(self._resize_grid), self._submit]
        return ops

    @staticmethod
    def _gen_color(c):
        def color(state, action):
            sel = action["selection"]
            if np.any(sel):
                state["grid"] = np.where(sel != 0, np.int8(c), state["grid"]).astype(np.int8)
        return color

    @staticmethod
    def _gen_flood_fill(c):
        def flood(state, action):
            sel = np.asarray(action["selection"], np.int8)
            if int(sel.sum()) != 1:
                return
            x, y = np.unravel_index(int(np.argmax(sel)), sel.shape)
            gh, gw = int(state["grid_dim"][0]), int(state["grid_dim"][1])
            if x >= gh or y >= gw:
                return
            g = state["grid"]
            col = g[x, y]
            seen = np.zeros((gh, gw), bool)
            stack = [(int(x), int(y))]
            while stack:  # the reference recurses (color.py:16-28); same visit set
                i, j = stack.pop()
                if i < 0 or j < 0 or i >= gh or j >= gw or seen[i, j] or g[i, j] != col:
                    continue
                seen[i, j] = True
                stack.extend(((i + 1, j), (i - 1, j), (i, j + 1), (i, j - 1)))
            g[:gh, :gw][seen] = c
        return flood

    def _gen_move(self, d):
        dx, dy = ((-1, 0), (1, 0), (0, 1), (0, -1))[d]

        def move(state, action):
            if not self._objsel(state, action["selection"]):
                return
            o = state["object_states"]
            o["object_pos"][:] = (_i8(int(o["object_pos"][0]) + dx), _i8(int(o["object_pos"][1]) + dy))
            self._place(state)
        return move

    def _gen_rotate(self, k):
        def rotate(state, action):
            if not self._objsel(state, action["selection"]):
                return
            o = state["object_states"]
            h, w = int(o["object_dim"][0]), int(o["object_dim"][1])
            x, y = int(o["object_pos"][0]), int(o["object_pos"][1])
            if h % 2 == w % 2:
                mod = 0
            else:
                o["rotation_parity"][0] = (int(o["rotation_parity"][0]) + k) % 2
                mod = 1 - int(o["rotation_parity"][0])
            o["object_pos"][:] = (x + (h - w) // 2 + mod, y + (w - h) // 2 + mod)
            o["object_dim"][:] = (w, h)
            for key in ("object", "object_sel"):
                t = np.rot90(o[key][:h, :w], k)
                o[key] = np.zeros_like(o[key])
                o[key][:w, :h] = t
            self._place(state)
        return rotate

    def _gen_flip(self, axis):
        def flip(state, action):
            if not self._objsel(state, action["selection"]):
                return
            o = state["object_states"]
            h, w = int(o["object_dim"][0]), int(o["object_dim"][1])
            for key in ("object", "object_sel"):
                t = o[key][:h, :w]
                o[key][:h, :w] = t[:, ::-1] if axis == 0 else t[::-1, :]
            self._place(state)
        return flip

    @staticmethod
    def _gen_copy(from_grid):
        def copy(state, action):
            sel = action["selection"]
            if not np.any(sel > 0):
                return
            x0, x1, y0, y1 = _bbox(sel != 0)
            sh, sw = (state["grid_dim"] if from_grid else state["input_dim"])
            if x1 > sh or y1 > sw:  # (sic: off by one)
                return
            src = state["grid"] if from_grid else state["input"]
            h, w = x1 - x0 + 1, y1 - y0 + 1
            state["clip"] = np.zeros_like(state["clip"])
            state["clip_dim"][:] = (h, w)
            s, m = src[x0:x1 + 1, y0:y1 + 1], sel[x0:x1 + 1, y0:y1 + 1]
            state["clip"][:h, :w] = np.where((s != 0) & (m != 0), s, 0)
        return copy

    @staticmethod
    def _paste(state, action):
        sel = action["selection"]
        if not np.any(sel > 0):
            return
        x0, _, y0, _ = _bbox(sel != 0)
        h, w = int(state["clip_dim"][0]), int(state["clip_dim"][1])
        if h == 0 or w == 0:
            return
        H, W = state["grid"].shape
        ex, ey = min(x0 + h, H), min(y0 + w, W)
        state["grid"][x0:ex, y0:ey] = state["clip"][:ex - x0, :ey - y0]

    @staticmethod
    def _copy_from_input(state, action):
        state["grid_dim"][:] = state["input_dim"]
        state["grid"] = state["input"].copy()

    @staticmethod
    def _reset_grid(state, action):
        state["grid"] = np.zeros_like(state["grid"])

    @staticmethod
    def _resize_grid(state, action):
        sel = action["selection"]
        if np.any(sel):
            x0, x1, y0, y1 = _bbox(sel != 0)
            state["grid"] = np.zeros_like(state["grid"])
            state["grid_dim"][:] = (x1 - x0 + 1, y1 - y0 + 1)

    def _grid_is_answer(self, state):
        gh, gw = int(state["grid_dim"][0]), int(state["grid_dim"][1])
        return self.answer.shape == (gh, gw) and bool(np.all(state["grid"][:gh, :gw] == self.answer))

    def _submit(self, state, action):
        if state["trials_remain"][0] != 0:
            state["trials_remain"][0] = _i8(int(state["trials_remain"][0]) - 1)
            self.submit_count += 1
            if self._grid_is_answer(state):
                state["terminated"][0] = 1
        if state["trials_remain"][0] == 0:
            state["terminated"][0] = 1

    # ---- step (A.5): the hot path ----------------------------------------------------------------------------
    def step(self, action):
        op = int(action["operation"])
        self.operations[op](self.state, action)
        reward = int(op == len(self.operations) - 1 and self._grid_is_answer(self.state))
        self.action_steps += 1
        info = {"steps": self.action_steps, "submit_count": self.submit_count}
        return self.state, reward, bool(self.state["terminated"][0]), False, info


def bbox_action(H, W, x1, y1, x2, y2, op):
    """BBoxWrapper-style action (5-tuple -> mask)."""
    sel = np.zeros((H, W), np.int8)
    sel[min(x1, x2):max(x1, x2) + 1, min(y1, y2):max(y1, y2) + 1] = 1
    return {"selection": sel, "operation": op}


def run_chunk(args):
    """Worker of the multi-process baseline: steps `n_envs` envs for `steps` C3-style random actions; returns env-steps done
    and the elapsed seconds of the stepping loop alone."""
    import time
    seed, n_envs, steps, H, W = args[:5]
    budget_s = args[5] if len(args) > 5 else None  # optional wall-clock budget: stop after the step during which it runs out
    rng = np.random.default_rng(seed)
    envs = []
    for _ in range(n_envs):
        e = NumpyO2ARCEnv(H, W, -1)
        ih, iw = rng.integers(1, H + 1), rng.integers(1, W + 1)
        g = rng.integers(0, 10, (ih, iw)).astype(np.int8)
        e.reset(g, g.copy())
        envs.append(e)
    bb = rng.integers(0, H, (steps, n_envs, 4))
    ops = rng.integers(0, 35, (steps, n_envs))
    t0 = time.perf_counter()
    done = 0
    for s in range(steps):
        for n, e in enumerate(envs):
            b = bb[s, n]
            e.step(bbox_action(H, W, int(b[0]), int(b[1]), int(b[2]), int(b[3]), int(ops[s, n])))
        done += n_envs
        if budget_s is not None and time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0
