"""AbstractARCEnv — host-side mirror of /root/reference/arcle/envs/base.py on top of the HIP step kernel.

Same constructor kwargs, `reset(seed, options)` option keys, obs dict keys / dtypes / shapes, `info` keys,
`create_operations()` plugin point, `op_names`, `transition`, `submit`, `reward` — but the state lives in HBM
(`EnvBatch`) and every transition is executed by libarcle_hip.so.  A single env (the Gymnasium API of the
reference) is simply a batch of one; `ARCVecEnv` (vec.py) is the batched front-end that the throughput
numbers are quoted on.
"""
from abc import ABCMeta, abstractmethod

import numpy as np
import torch

from .. import actions, spaces
from ..engine import EnvBatch, ST_BAD_OP, ST_ROTATE_DOMAIN, STEP_FLAT_OBS, STEP_RESET_ON_SUBMIT
from ..loaders import Loader


class AbstractARCEnv(spaces.Env, metaclass=ABCMeta):
    """Abstract ARC environment (base.py:15-66).  Subclasses define KIND, STATE_KEYS and create_operations()."""

    metadata = {"render_modes": [], "render_fps": 5}
    KIND = "raw"

    def __init__(self, data_loader: Loader, max_grid_size, colors, max_trial=-1, render_mode=None, render_size=None,
                 device=None):
        if render_mode is not None:
            raise NotImplementedError("rendering is outside the ported hot path (SURVEY.md §2 row 1)")
        self.loader = data_loader
        self.H, self.W = int(max_grid_size[0]), int(max_grid_size[1])
        self.colors = colors
        self.max_trial = max_trial
        self.render_mode = render_mode
        self.render_size = render_size
        self.rendering = None
        self.device = device
        # op table (plugin point, base.py:61,140-142)
        self.operations = self.create_operations()
        self._descs = actions.table_descs(self.operations)
        self.observation_space = self.create_state_space()
        self.action_space = self.create_action_space(len(self.operations))
        self.op_names = ["".join(map(str.capitalize, op.__name__.split("_"))) for op in self.operations]  # base.py:66
        self._batch = None
        self._scratch = None
        self.current_state = None
        self.input_ = self.answer = self.description = None
        self.last_action = self.last_action_op = None
        self.last_reward = 0
        self.action_steps = 0
        self.submit_count = 0
        self.truncated = False

    # ---- device plumbing -----------------------------------------------------------------------
    def _new_batch(self, n):
        b = EnvBatch(n, self.H, self.W, self.max_trial, self.KIND, self.device)
        b.set_op_table(self._descs)
        return b

    @property
    def batch(self):
        if self._batch is None:
            self._batch = self._new_batch(1)
        return self._batch

    # ---- one-env fast path: the whole state crosses PCIe as ONE flattened row (written by the step kernel itself) ---------
    def _io(self):
        """Persistent staging buffers of the single env: pinned host + device tensors for the action, the flattened
        observation row (arcle_set_flat_output) and the step outputs."""
        if getattr(self, "_io_bufs", None) is None:
            b = self.batch
            flat = b.set_flat_output(False)
            pin = torch.cuda.is_available()
            mk = lambda shape, dt: torch.zeros(shape, dtype=dt, pin_memory=pin)  # noqa: E731
            self._io_bufs = dict(flat=flat, h_flat=mk(tuple(flat.shape), torch.int8), h_sel=mk((1, self.H, self.W), torch.int8),
                                 d_sel=torch.zeros((1, self.H, self.W), dtype=torch.int8, device=b.device), h_op=mk((1,), torch.int32),
                                 d_op=torch.zeros(1, dtype=torch.int32, device=b.device), h_cnt=mk((1, 2), torch.int32),
                                 h_reward=mk((1,), torch.int32))
        return self._io_bufs

    def _flat_layout(self):
        """(key path, length) of the full flattened row in the order arcle_flatten_obs writes it (FlattenObservation's sorted
        keys, arcle_amd/csrc/arcle_wave.h `flat_row`)."""
        P, kind = self.H * self.W, self.KIND
        lay = []
        if kind != "raw":
            lay += [(("clip",), P), (("clip_dim",), 2)]
        lay += [(("grid",), P), (("grid_dim",), 2), (("input",), P), (("input_dim",), 2)]
        if kind == "o2arc":
            o = "object_states"
            lay += [((o, "active"), 1), ((o, "background"), P), ((o, "object"), P), ((o, "object_dim"), 2), ((o, "object_pos"), 2),
                    ((o, "object_sel"), P), ((o, "rotation_parity"), 1), (("selected",), P)]
        lay += [(("terminated",), 1), (("trials_remain",), 1)]
        return lay

    def _state_from_row(self, row):
        """The reference's obs dict (fresh numpy int8 arrays) from one flattened row (numpy int8 [L])."""
        P, st, off = self.H * self.W, {}, 0
        for path, n in self._flat_layout():
            v = row[off:off + n].copy()
            off += n
            if n == P:
                v = v.reshape(self.H, self.W)
            d = st
            for k in path[:-1]:
                d = d.setdefault(k, {})
            d[path[-1]] = v
        return st

    def _fetch_state(self, b, launch=True):
        """Current state as the obs dict through one device->host copy of the flattened row."""
        io = self._io()
        if launch:
            b.flat_obs(out=b._flat_buf, filtered=False)
        io["h_flat"].copy_(io["flat"], non_blocking=True)
        torch.cuda.synchronize(b.device)
        return self._state_from_row(io["h_flat"][0].numpy())

    @staticmethod
    def _state_from_device(b, n=0):
        """Builds the reference's obs dict (numpy int8 arrays) for env n of batch b."""
        rec = b.rec[n].cpu().numpy()
        f = lambda name: rec[slice(*_span(name))].copy()  # noqa: E731
        st = {"trials_remain": f("trials_remain"), "terminated": f("terminated"),
              "input": b.plane("input")[n].cpu().numpy(), "input_dim": f("input_dim"),
              "grid": b.plane("grid")[n].cpu().numpy(), "grid_dim": f("grid_dim")}
        if "clip" in b.planes:
            st["clip"] = b.plane("clip")[n].cpu().numpy()
            st["clip_dim"] = f("clip_dim")
        if "selected" in b.planes:
            st["selected"] = b.plane("selected")[n].cpu().numpy()
            st["object_states"] = {
                "active": f("active"), "object": b.plane("object")[n].cpu().numpy(),
                "object_sel": b.plane("object_sel")[n].cpu().numpy(), "object_dim": f("object_dim"),
                "object_pos": f("object_pos"), "background": b.plane("background")[n].cpu().numpy(),
                "rotation_parity": f("rotation_parity")}
        return st

    @staticmethod
    def _state_to_device(b, state, n=0):
        dev = b.device
        put = lambda name, arr: b.plane(name)[n].copy_(torch.as_tensor(np.asarray(arr, np.int8), device=dev))  # noqa: E731
        put("input", state["input"])
        put("grid", state["grid"])
        rec = b.rec[n].cpu().numpy().copy()
        flat = dict(state)
        if "object_states" in state:
            flat.update(state["object_states"])
            for k in ("selected", "clip", "object", "object_sel", "background"):
                put(k, flat[k])
        elif "clip" in state:
            put("clip", state["clip"])
        for k in ("input_dim", "grid_dim", "clip_dim", "object_dim", "object_pos", "trials_remain", "terminated",
                  "active", "rotation_parity"):
            if k in flat:
                lo, hi = _span(k)
                rec[lo:hi] = np.asarray(flat[k], np.int8)
        b.rec[n].copy_(torch.as_tensor(rec, device=dev))

    # ---- Gymnasium API -------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """base.py:69-118 — same option keys: prob_index, subprob_index, adaptation, reset_on_submit."""
        self.truncated = False
        self.submit_count = 0
        self.last_action = self.last_action_op = None
        self.last_reward = 0
        self.action_steps = 0
        self.prob_index = self.subprob_index = None
        self.adaptation = True
        self.reset_on_submit = False
        self.options = options
        if options is not None:
            self.prob_index = options.get("prob_index")
            self.subprob_index = options.get("subprob_index")
            _ad = options.get("adaptation")
            self.adaptation = True if _ad is None else bool(_ad)
            _ros = options.get("reset_on_submit")
            self.reset_on_submit = False if _ros is None else _ros
        if spaces.HAVE_GYMNASIUM:
            super().reset(seed=seed)  # seeds np_random like the reference's super().reset (base.py:70)
        ex_in, ex_out, tt_in, tt_out, desc = self.loader.pick(data_index=self.prob_index)
        src_in, src_out = (ex_in, ex_out) if self.adaptation else (tt_in, tt_out)
        if self.subprob_index is None:
            self.subprob_index = np.random.randint(0, len(src_in))  # global np.random, as base.py:99,104
        self.input_ = src_in[self.subprob_index]
        self.answer = src_out[self.subprob_index]
        self.description = desc
        b = self.batch
        b.set_tasks([self.input_], [self.answer])
        b.reset()
        self.current_state = self._fetch_state(b)
        self.info = self.init_info()
        if self.render_mode:
            self.render()
        return self.current_state, self.info

    def init_info(self):
        isize, osize = self.input_.shape, self.answer.shape
        return {"input": np.pad(self.input_, [(0, self.H - isize[0]), (0, self.W - isize[1])], constant_values=0),
                "input_dim": isize,
                "answer": np.pad(self.answer, [(0, self.H - osize[0]), (0, self.W - osize[1])], constant_values=0),
                "answer_dim": osize}

    def _step_flags(self):
        return STEP_RESET_ON_SUBMIT if self.reset_on_submit else 0

    def _device_step(self, b, action):
        op = int(action["operation"])
        if not -len(self.operations) <= op < len(self.operations):
            raise IndexError("list index out of range")  # what self.operations[op] raises in the reference
        op %= len(self.operations)  # (a negative index counts from the end, like the Python list of the reference)
        sel = np.asarray(action["selection"])
        if sel.shape != (self.H, self.W):
            raise ValueError(f"selection must have shape {(self.H, self.W)}")
        fn = self.operations[op]
        if not isinstance(fn, actions.Operation) and not actions.is_submit(fn):
            # an arbitrary Python callable in the table (base.py:140-142 allows it; agents/wrapper.py:53-57): applied on the
            # host to the fetched state, written back, and the step's bookkeeping done by a device no-op slot
            state = self._state_from_device(b)
            fn(state, action)
            self._state_to_device(b, state)
        # action in: pinned staging -> device (asynchronous); step with the fused observation row; row, counters and reward out
        # (asynchronous); the status read synchronises the stream
        if b is not self._batch:  # the scratch env of transition(): plain path, the caller fetches the state itself
            sel_t = torch.as_tensor(sel.astype(np.int8, copy=False), device=b.device).reshape(1, self.H, self.W)
            reward, term = b.step_mask(sel_t, torch.tensor([op], dtype=torch.int32, device=b.device), self._step_flags())
            st = b.status()
            if st & ST_ROTATE_DOMAIN:
                raise ValueError("Rotate/Flip outside its domain (the reference raises here too: object.py:45 / int8 overflow)")
            if st & ST_BAD_OP:
                raise IndexError("list index out of range")
            return int(reward[0]), bool(term[0])
        io = self._io()
        io["h_sel"][0].copy_(torch.from_numpy(np.ascontiguousarray(sel.astype(np.int8, copy=False))))
        io["h_op"][0] = op
        io["d_sel"].copy_(io["h_sel"], non_blocking=True)
        io["d_op"].copy_(io["h_op"], non_blocking=True)
        reward, term = b.step_mask(io["d_sel"], io["d_op"], self._step_flags() | STEP_FLAT_OBS)
        io["h_flat"].copy_(io["flat"], non_blocking=True)
        io["h_cnt"].copy_(b.cnt, non_blocking=True)
        io["h_reward"].copy_(reward, non_blocking=True)
        st = b.status()
        if st & ST_ROTATE_DOMAIN:
            raise ValueError("Rotate/Flip outside its domain (the reference raises here too: object.py:45 / int8 overflow)")
        if st & ST_BAD_OP:
            raise IndexError("list index out of range")
        self._row_ready = True
        return int(io["h_reward"][0]), bool(io["h_flat"][0, -2] != 0)  # (terminated is the row's last-but-one byte)

    def step(self, action):
        """o2arcenv.py:130-147 / arcenv.py:60-76,155-172."""
        b = self.batch
        if type(self).transition is not AbstractARCEnv.transition:
            # a subclass overrides transition() (the reference's step calls self.transition(self.current_state, action),
            # o2arcenv.py:134): run it on the state dict, then write the dict back and do the bookkeeping here
            self.transition(self.current_state, action)
            self._state_to_device(b, self.current_state)
            self.action_steps += 1
            b.cnt[0, 0] = self.action_steps
            b.cnt[0, 1] = self.submit_count
            self.last_action_op = int(action["operation"])
            self.last_action = action
            reward = self.reward(self.current_state)
            term = bool(self.current_state["terminated"][0])
        else:
            dev_reward, term = self._device_step(b, action)
            self.last_action_op = int(action["operation"]) % len(self.operations)
            self.last_action = action
            io = self._io()
            self.current_state = self._state_from_row(io["h_flat"][0].numpy())  # (copied out by _device_step)
            self.action_steps, self.submit_count = int(io["h_cnt"][0, 0]), int(io["h_cnt"][0, 1])
            # a subclass that overrides reward() (e.g. the dense reward of agents/env.py:44-58) is evaluated on the host;
            # so is the reward of a host-applied last op
            host_reward = type(self).reward is not AbstractARCEnv.reward or not isinstance(
                self.operations[self.last_action_op], actions.Operation) and not actions.is_submit(self.operations[self.last_action_op])
            reward = self.reward(self.current_state) if host_reward else dev_reward
        self.last_reward = reward
        self.info["steps"] = self.action_steps
        if "submit_count" in self.info:
            self.info["submit_count"] = self.submit_count
        return self.current_state, reward, term, self.truncated, self.info

    def transition(self, state, action):
        """o2arcenv.py:149-151 — applies one operation to `state` IN PLACE (README usage:
        `env.transition(deepcopy(state), action)`).  The given dict is uploaded into a scratch env,
        stepped by the kernel and written back.  Like the reference, a Submit routed through here counts
        (`self.submit_count`, base.py:175); `action_steps` does not move."""
        if self._scratch is None:
            self._scratch = self._new_batch(1)
        s = self._scratch
        s.set_tasks([self.input_], [self.answer])
        self._state_to_device(s, state)
        s.cnt.zero_()
        self._device_step(s, action)
        self.submit_count += int(s.cnt[0, 1])
        new = self._state_from_device(s)
        for k, v in new.items():
            if k == "object_states":
                state.setdefault("object_states", {}).update(v)
            else:
                state[k] = v

    def submit(self, state, action):
        """base.py:172-183 as an operation on `state` (the table slot `self.submit` maps to the device op)."""
        a = dict(action)
        a["operation"] = next(i for i, d in enumerate(self._descs) if d & 0xFF == actions.OP_SUBMIT)
        self.transition(state, a)

    def reward(self, state):
        """o2arcenv.py:121-128: 1 iff the last action was the LAST op and grid[:h,:w] == answer."""
        if not self.last_action_op == len(self.operations) - 1:
            return 0
        if tuple(state["grid_dim"]) == self.answer.shape:
            h, w = self.answer.shape
            if np.all(state["grid"][0:h, 0:w] == self.answer):
                return 1
        return 0

    # ---- spaces (base.py:121-138) ----------------------------------------------------------------
    def create_state_space(self):
        S = spaces
        return S.Dict({
            "trials_remain": S.Box(-1, self.max_trial, shape=(1,), dtype=np.int8),
            "terminated": S.MultiBinary(1),
            "input": S.Box(0, self.colors, (self.H, self.W), dtype=np.int8),
            "input_dim": S.Box(low=np.array([1, 1]), high=np.array([self.H, self.W]), dtype=np.int8),
            "grid": S.Box(0, self.colors, (self.H, self.W), dtype=np.int8),
            "grid_dim": S.Box(low=np.array([1, 1]), high=np.array([self.H, self.W]), dtype=np.int8),
        })

    def create_action_space(self, action_count):
        return spaces.Dict({"selection": spaces.Box(0, 1, (self.H, self.W), dtype=np.int8),
                            "operation": spaces.Discrete(action_count)})

    @abstractmethod
    def create_operations(self):
        pass

    def render(self):
        """Rendering (base.py:185-224) is outside the ported path: no-op."""


def _span(name):
    from ..engine import REC_FIELDS
    off, n = REC_FIELDS[name]
    return off, off + n
