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
from ..engine import EnvBatch, ST_BAD_OP, ST_ROTATE_DOMAIN, STEP_FLAT_OBS, STEP_RESET_ON_SUBMIT, check_grid_size
from ..loaders import Loader


class AbstractARCEnv(spaces.Env, metaclass=ABCMeta):
    """Abstract ARC environment (base.py:15-66).  Subclasses define KIND, STATE_KEYS and create_operations()."""

    metadata = {"render_modes": [], "render_fps": 5}
    # Observations: the reference hands out ONE state dict per env and mutates its arrays in place from step to step (np.copyto /
    # state['grid'][:, :] = ..., actions/object.py:80-165, critical.py:17-65), so an observation kept from an earlier step changes under
    # later ones.  Same here by default: the arrays of the dict step() returns are views of the pinned host row the step kernel writes
    # (zero copies, nothing rebuilt per step).  True: an independent copy of the row per step (round 4's behaviour, +6 us per step).
    copy_observations = False
    KIND = "raw"

    def __init__(self, data_loader: Loader, max_grid_size, colors, max_trial=-1, render_mode=None, render_size=None,
                 device=None):
        if render_mode is not None:
            raise NotImplementedError("rendering is outside the ported hot path (SURVEY.md §2 row 1)")
        self.loader = data_loader
        self.H, self.W = int(max_grid_size[0]), int(max_grid_size[1])
        check_grid_size(self.H, self.W)  # (sides <= 127; planes of more than 1024 cells run on the workgroup-per-env kernels)
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

    # ---- one-env fast path: ONE launch and ONE synchronisation per step(), no copies -----------------------------------------
    # The action (selection mask + op index) is written into PINNED HOST memory the kernel reads directly, and the step kernel
    # writes the flattened state row — with the step outputs (reward, counters, terminated, per-env status) in the row's 16-byte
    # tail — straight into pinned host memory as well (arcle_set_flat_output_ex).  After the stream synchronisation everything
    # step() returns is on the host.  (Round 2: 2 H2D copies + step + status kernel + 3 D2H copies + 2 synchronisations.)
    def _io(self):
        if getattr(self, "_io_bufs", None) is None:
            b = self.batch
            flat = b.set_flat_output(False, tail=True, host=True)
            P = self.H * self.W
            act = torch.zeros(((P + 3) & ~3) + 4, dtype=torch.int8).pin_memory()  # selection bytes, then the op as int32
            self._io_bufs = dict(flat=flat, row=flat[0].numpy(), tail=b.flat_tail[0].numpy(), tail_u8=b.flat_tail[0].numpy().view(np.uint8),
                                 act=act, sel=act[:P].numpy(),
                                 op=act[(P + 3) & ~3:].view(torch.int32).numpy(), sel_ptr=act.data_ptr(),
                                 op_ptr=act.data_ptr() + ((P + 3) & ~3), L=flat.shape[1])
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

    def _row_plan(self):
        if getattr(self, "_plan", None) is None:
            P, off, plan = self.H * self.W, 0, []
            for path, n in self._flat_layout():
                plan.append((path, off, off + n, (self.H, self.W) if n == P else None))
                off += n
            self._plan, self._row_len = plan, off
        return self._plan

    def _state_from_row(self, row, live=False):
        """The reference's obs dict from one flattened row.  live (the env's own pinned row, copy_observations False): a dict whose
        arrays ARE the row — the kernel rewrites them under the caller.  Otherwise fresh numpy int8 arrays, independent of the
        staging buffer: ONE copy of the row, every key a view into that copy."""
        plan = self._row_plan()
        buf = row[:self._row_len] if live and not self.copy_observations else row[:self._row_len].copy()
        st = {}
        for path, lo, hi, shape in plan:
            v = buf[lo:hi]
            if shape is not None:
                v.shape = shape
            if len(path) == 1:
                st[path[0]] = v
            else:
                st.setdefault(path[0], {})[path[1]] = v
        return st

    def _new_live_state(self):
        """A NEW state dict over the pinned row — one per episode, like the reference's init_state (base.py:156-167: reset, and Submit
        under reset_on_submit, bind self.current_state to a fresh dict; between them step() hands out the same dict every time).  The
        views are remembered so that a key rebound behind the env's back (a reference-style op doing `state['grid'] = ...`,
        color.py:73) can be pointed at the row again."""
        st = self._state_from_row(self._io()["row"], live=True)
        self._live_views = None
        if not self.copy_observations:
            self._live_views = [(path, st[path[0]] if len(path) == 1 else st[path[0]][path[1]]) for path, _, _, _ in self._row_plan()]
            self._live_nested = st.get("object_states")
        self.current_state = st
        return st

    def _live_state(self):
        """The state dict after a device step.  Live views: the episode's own dict, keys somebody rebound pointed at the row again;
        copy_observations: a fresh dict over an independent copy of the row."""
        if self._live_views is None:
            self.current_state = self._state_from_row(self._io()["row"])
            return self.current_state
        st = self.current_state
        if self._live_nested is not None and st.get("object_states") is not self._live_nested:
            st["object_states"] = self._live_nested
        for path, v in self._live_views:
            d = st if len(path) == 1 else self._live_nested
            if d.get(path[-1]) is not v:
                d[path[-1]] = v
        return st

    def _detach_live_state(self):
        """Before the pinned row is rewritten for a NEW episode: the previous episode's dict (somebody may have kept it — the
        reference's reset leaves the old dict alone) gets its own memory."""
        st = self.current_state
        if st is None or getattr(self, "_live_views", None) is None:
            return
        buf = self._io()["row"][:self._row_len].copy()
        for path, lo, hi, shape in self._row_plan():
            d = st if len(path) == 1 else st.get(path[0])
            if isinstance(d, dict) and any(d.get(path[-1]) is v for p2, v in self._live_views if p2 == path):
                d[path[-1]] = buf[lo:hi].reshape(shape) if shape is not None else buf[lo:hi]
        self._live_views = None

    def _fetch_state(self, b):
        """Current state as a NEW obs dict: one flatten launch that writes the row into pinned host memory, one synchronisation."""
        self._detach_live_state()
        io = self._io()
        buf = b._flat_buf
        b._check(b.L.arcle_flatten_obs(b._h, buf.data_ptr(), buf.shape[1], 0, b._stream()), "arcle_flatten_obs")
        b.sync()
        return self._new_live_state()

    def _row_from_state(self, state, out):
        """Inverse of _state_from_row: writes the obs dict `state` into the numpy int8 row `out` (full layout)."""
        for path, lo, hi, _ in self._row_plan():
            v = state[path[0]] if len(path) == 1 else state[path[0]][path[1]]
            out[lo:hi] = v.ravel() if type(v) is np.ndarray and v.dtype == np.int8 else np.asarray(v, np.int8).reshape(-1)
        return self._row_len

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
        b.invalidate()  # (the planes were written behind the library's back)

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
        self.current_state = self._fetch_state(b)  # (a new dict per episode)
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

    def _check_action(self, action):
        """(operation as given, its table slot, selection).  A negative index counts from the end like the Python list of the
        reference (`self.operations[op]`, o2arcenv.py:149-151); the un-normalised value is what `last_action_op` keeps."""
        raw = int(action["operation"])
        n = len(self.operations)
        if not -n <= raw < n:
            raise IndexError("list index out of range")  # what self.operations[op] raises in the reference
        sel = np.asarray(action["selection"])
        if sel.shape != (self.H, self.W):
            raise ValueError(f"selection must have shape {(self.H, self.W)}")
        return raw, raw % n, sel

    def _put_selection(self, sel, out):
        """Selection of any dtype / memory layout -> the int8 mask the kernel reads (`out`: flat pinned view of H*W bytes).  int8 and
        bool go in as they are; other dtypes (uint8, wider ints, floats) are accepted when every value is exactly representable as
        int8 — then truthiness, `> 0`, the sum and the arg-max the operations take of the mask (color.py:71-99, object.py:68-88) are
        what the reference computes on the original.  Anything else (200 in a uint8 mask, 0.5, NaN) is refused by name: the int8
        kernel cannot reproduce what NumPy would do with it."""
        out2 = out.reshape(self.H, self.W)
        with np.errstate(invalid="ignore"):  # (a NaN casts to some int8; the comparison below refuses it)
            np.copyto(out2, sel, casting="unsafe")
        if sel.dtype != np.int8 and sel.dtype != np.bool_ and not np.array_equal(out2, sel):
            raise ValueError(f"selection of dtype {sel.dtype} holds values that are not representable as int8 "
                             "(the action space is Box(0, 1, int8), base.py:135)")

    @staticmethod
    def _raise_status(st):
        if st & ST_ROTATE_DOMAIN:
            raise ValueError("Rotate/Flip outside its domain (the reference raises here too: object.py:45 / int8 overflow)")
        if st & ST_BAD_OP:
            raise IndexError("list index out of range")

    def _device_step(self, b, op, sel, action):
        fn = self.operations[op]
        if not isinstance(fn, actions.Operation) and not actions.is_submit(fn):
            # an arbitrary Python callable in the table (base.py:140-142 allows it; agents/wrapper.py:53-57): applied on the
            # host to the fetched state, written back, and the step's bookkeeping done by a device no-op slot
            state = self._state_from_device(b)
            fn(state, action)
            self._state_to_device(b, state)
        io = self._io()
        self._put_selection(sel, io["sel"])
        io["op"][0] = op
        st = b._stream()
        seq = b.next_seq(io["tail_u8"])
        b._check(b.L.arcle_step_mask(b._h, io["sel_ptr"], io["op_ptr"], b._reward_ptr, b._term_ptr,
                                     self._step_flags() | STEP_FLAT_OBS, st), "arcle_step_mask")
        b.wait_tail(io["tail_u8"], seq, st)  # the one wait of the step: the kernel's own completion signal in the pinned row's tail
        tail = io["tail"]
        status = (int(tail[3]) >> 16) & 0xFF
        if status:
            b.status()  # (clears the handle's sticky word as well)
            self._raise_status(status)
        return int(tail[0]), bool(int(tail[3]) & 0xFF)

    # RawARCEnv.step records the action before it applies the op (arcenv.py:62-68), the other classes after (o2arcenv.py:132-136,
    # arcenv.py:157-161): it only shows when the op raises
    _RECORD_ACTION_FIRST = False

    def step(self, action):
        """o2arcenv.py:130-147 / arcenv.py:60-76,155-172."""
        b = self.batch
        if self._RECORD_ACTION_FIRST:
            self.last_action_op, self.last_action = int(action["operation"]), action
        if type(self).transition is not AbstractARCEnv.transition:
            # a subclass overrides transition() (the reference's step calls self.transition(self.current_state, action),
            # o2arcenv.py:134): run it on the state dict, then write the dict back and do the bookkeeping here
            self._in_step = True
            try:
                self.transition(self.current_state, action)
            finally:
                self._in_step = False
            self._state_to_device(b, self.current_state)
            self.action_steps += 1
            b.cnt[0, 0] = self.action_steps
            b.cnt[0, 1] = self.submit_count
            self.last_action_op = int(action["operation"])
            self.last_action = action
            reward = self.reward(self.current_state)
            term = bool(self.current_state["terminated"][0])
        else:
            raw, op, sel = self._check_action(action)
            dev_reward, term = self._device_step(b, op, sel, action)
            self.last_action_op = raw
            self.last_action = action
            io = self._io()
            submits = int(io["tail"][2])
            resubmitted = self.reset_on_submit and submits != self.submit_count
            self.action_steps, self.submit_count = int(io["tail"][1]), submits
            if resubmitted and not self._step_flags():
                self._reinit_state(b)   # (RawARCEnv: the kernel stepped without RESET_ON_SUBMIT, see its _step_flags)
            elif resubmitted:
                self._new_live_state()  # Submit re-initialised the state: a new dict, like init_state (base.py:179-180)
            else:
                self._live_state()      # (the kernel wrote the row into pinned host memory)
            # evaluated on the host: the reward of a subclass that overrides reward() (e.g. the dense one of agents/env.py:44-58), of a
            # host-applied last op, and of an op given by a negative index (`last_action_op == len(ops) - 1` is false for -1)
            host_reward = raw != op or type(self).reward is not AbstractARCEnv.reward or not isinstance(
                self.operations[op], actions.Operation) and not actions.is_submit(self.operations[op])
            reward = self.reward(self.current_state) if host_reward else dev_reward
        self.last_reward = reward
        self.info["steps"] = self.action_steps
        if "submit_count" in self.info:
            self.info["submit_count"] = self.submit_count
        return self.current_state, reward, term, self.truncated, self.info

    def _tio(self):
        """Pinned host staging of transition(): state row in, action in, state row + tail out — the kernel
        (arcle_transition_rows) reads and writes all of it in place, nothing is copied and no resident env is touched."""
        if getattr(self, "_tio_bufs", None) is None:
            b = self.batch
            L, P = b.state_row_size(), self.H * self.W
            stride = ((L + 15) & ~15) + 16
            rin = torch.zeros((1, stride), dtype=torch.int8).pin_memory()
            rout = rin  # IN PLACE: the kernel fetches only the planes the op reads and rewrites only the ones it changed
            act = torch.zeros(((P + 3) & ~3) + 4, dtype=torch.int8).pin_memory()
            self._tio_bufs = dict(rin=rin, rout=rout, act=act, rin_np=rin[0].numpy(), rout_np=rout[0].numpy(), sel=act[:P].numpy(),
                                  op=act[(P + 3) & ~3:].view(torch.int32).numpy(), tail=rout[0, stride - 16:].view(torch.int32).numpy(),
                                  tail_u8=rout[0, stride - 16:].numpy().view(np.uint8),
                                  stride=stride, L=L, op_off=(P + 3) & ~3)
        return self._tio_bufs

    def transition(self, state, action):
        """o2arcenv.py:149-151 — applies one operation to `state` IN PLACE (README usage: `env.transition(deepcopy(state), action)`).
        One launch of the stateless row kernel: the dict becomes a state row in pinned host memory, arcle_transition_rows applies the
        op, the result is written back into the dict's own arrays.  Like the reference, a Submit routed through here counts
        (`self.submit_count`, base.py:175) and, under reset_on_submit, re-initialises the ENV's state (base.py:179-180:
        `self.init_state`, a new `current_state` dict) while `state` itself only loses a trial; `action_steps` and `last_action_op`
        do not move.  `state is env.current_state`: the env's resident device state follows, as the reference's one dict does.
        (Planning over many states at once: ARCVecEnv.transition — same kernel, any number of rows per launch.)"""
        _, op, sel = self._check_action(action)
        fn = self.operations[op]
        if not isinstance(fn, actions.Operation) and not actions.is_submit(fn):
            fn(state, action)  # a host callable in the table: it IS the transition
            if state is self.current_state and not getattr(self, "_in_step", False):
                self._state_to_device(self.batch, state)
            return
        b = self.batch
        t = self._tio()
        self._row_from_state(state, t["rin_np"])
        self._put_selection(sel, t["sel"])
        t["op"][0] = op
        ap = t["act"].data_ptr()
        st = b._stream()
        seq = b.next_seq(t["tail_u8"])
        # (RESET_ON_SUBMIT is not passed: the row is the caller's `state`, which the reference does not re-initialise)
        b._check(b.L.arcle_transition_rows(b._h, 1, t["rin"].data_ptr(), t["stride"], 0, ap, ap + t["op_off"], None,
                                           t["rout"].data_ptr(), t["stride"], 1, b._reward_ptr, b._term_ptr, 0, st),
                 "arcle_transition_rows")
        b.wait_tail(t["tail_u8"], seq, st)
        status = (int(t["tail"][3]) >> 16) & 0xFF
        if status:
            b.status()
            self._raise_status(status)
        submitted = int(t["tail"][2])
        if submitted:
            self.submit_count += submitted
            b.cnt[0, 1] = self.submit_count  # (the env's counter, whatever state the Submit ran on: the next step() reports it)
        live = state is self.current_state and not getattr(self, "_in_step", False)
        row = t["rout_np"]
        for path, lo, hi, shape in self._row_plan():
            d = state if len(path) == 1 else state.setdefault(path[0], {})
            cur = d.get(path[-1])
            v = row[lo:hi]
            if type(cur) is np.ndarray and cur.size == hi - lo and cur.flags.writeable:
                np.copyto(cur, v.reshape(cur.shape), casting="unsafe")  # in place: the dict's arrays stay the caller's (and the env's)
            else:
                d[path[-1]] = v.reshape(shape).copy() if shape is not None else v.copy()
        if submitted and self.reset_on_submit:
            self._reinit_state(b)  # (a live `state` is orphaned by the new dict and keeps what was just written, in its own memory)
        elif live:
            b.set_state_rows(t["rout"])
            b.sync()

    def _reinit_state(self, b):
        """init_state(self.input_) in the middle of an episode (base.py:179-180): the env's state as reset() leaves it — a NEW dict —,
        counters kept."""
        b.reset()
        b.cnt[0, 0] = self.action_steps
        b.cnt[0, 1] = self.submit_count
        self.current_state = self._fetch_state(b)

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
