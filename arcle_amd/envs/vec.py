"""ARCVecEnv — N independent ARCLE envs stepped by ONE kernel launch per step (the hot path the
throughput numbers are quoted on).  Observations are zero-copy torch views of the device state (like the
reference, `obs` IS the live state — o2arcenv.py:147), actions are device tensors:

    venv = ARCVecEnv(O2ARCv2Env, num_envs=8192, data_loader=loader, max_grid_size=(30, 30))
    obs, info = venv.reset()
    obs, reward, terminated, truncated, info = venv.step_bbox(bbox_i32[N,4], op_i32[N])     # BBoxWrapper form
    obs, reward, terminated, truncated, info = venv.step_point(xy_i32[N,2], op_i32[N])      # PointWrapper form
    obs, reward, terminated, truncated, info = venv.step({"selection": m[N,H,W], "operation": op[N]})

Everything a rollout needs stays on the device: the task table (Loader.parse's output, uploaded once), the task choice
at reset / auto-reset (keyed by the GLOBAL env id, so a sharded batch walks the same tasks — arcle_amd/sampling.py), the
research env's epilogues (agents/env.py: dense reward, colour-permutation + rot90 augmentation; agents/train.py:67
TimeLimit truncation).
"""
import numpy as np
import torch

from .. import actions
from ..engine import (AUG_PERMUTE, AUG_ROT90, EnvBatch, STEP_AUTORESET, STEP_DENSE, STEP_PACK_OBS, STEP_RESAMPLE, STEP_RESET_ON_SUBMIT,
                      STEP_TRUNCATE, ST_BAD_OP, ST_BAD_SELECTION, ST_BAD_TASK, ST_ROTATE_DOMAIN)


def _table_of(env_cls):
    """The class's operation table: `create_operations()` is the reference's plugin point (base.py:140-142), so an
    override of it is honoured even though no env instance (and no device batch of one) is built here."""
    probe = object.__new__(env_cls)
    return list(probe.create_operations())


class _LazyInfo(dict):
    """A dict whose registered entries are computed on first access (d[k], d.get(k), `k in d`, iteration all see them)."""

    def __init__(self, base):
        super().__init__(base)
        self._thunks = {}

    def lazy(self, key, thunk):
        self._thunks[key] = thunk

    def __setitem__(self, key, value):
        self._thunks.pop(key, None)  # an explicit assignment replaces a pending entry
        dict.__setitem__(self, key, value)

    def _force(self, key=None):
        for k in ([key] if key is not None else list(self._thunks)):
            if k in self._thunks:
                dict.__setitem__(self, k, self._thunks.pop(k)())

    def __missing__(self, key):
        if key in self._thunks:
            self._force(key)
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        self._force(key)
        return dict.get(self, key, default)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._thunks

    def keys(self):
        self._force()
        return dict.keys(self)

    def items(self):
        self._force()
        return dict.items(self)

    def values(self):
        self._force()
        return dict.values(self)

    def __iter__(self):
        self._force()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self._thunks)


class ARCVecEnv:
    def __init__(self, env_cls, num_envs, data_loader=None, max_grid_size=(30, 30), colors=10, max_trial=None,
                 device=None, autoreset=False, operations=None, rng=None, seed=None, env_base=0,
                 max_episode_steps=None, dense_reward=False, augment=()):
        """env_cls: RawARCEnv / ARCEnv / O2ARCv2Env or a subclass (its `create_operations` and KIND define the op table
        and the state planes); `operations` overrides the table.
        autoreset: False | True (Gymnasium next-step autoreset onto the SAME task, inside the step kernel) |
                   "resample" (the same, onto a NEW task drawn on the device from the loader's tasks).
        seed / env_base: key of the device-side task draws (global env id = env_base + local index).
        max_episode_steps: TimeLimit — `truncated` turns True once an env has taken that many steps (agents/train.py:67).
        dense_reward: the research env's reward, sparse*100 - 1 + correct/total (agents/env.py:44-58), as float32.
        augment: subset of ("permute", "rot90") — task augmentation at every (re)start (agents/env.py:31-42).
        Everything `reset` / `step_*` return lives on the device and is a VIEW of this env's buffers (obs planes, reward,
        terminated, truncated, info entries): the next step overwrites them in place — copy what must outlive it."""
        self.env_cls, self.N = env_cls, int(num_envs)
        self.H, self.W = int(max_grid_size[0]), int(max_grid_size[1])
        self.colors = colors
        if max_trial is None:
            max_trial = 3 if env_cls.KIND == "arc" else -1  # the classes' defaults (arcenv.py:79, o2arcenv.py:14)
        self.max_trial = max_trial
        self.loader = data_loader
        self.operations = list(operations) if operations is not None else _table_of(env_cls)
        self.op_names = ["".join(map(str.capitalize, op.__name__.split("_"))) for op in self.operations]
        self._host_slots = actions.host_slots(self.operations)
        self.batch = EnvBatch(self.N, self.H, self.W, max_trial, env_cls.KIND, device)
        self.batch.set_op_table(actions.table_descs(self.operations))
        self.device = self.batch.device
        self.rng = rng if rng is not None else np.random.default_rng(seed)
        self.seed = int(seed) if seed is not None else int(self.rng.integers(0, 2**63))
        self.env_base = int(env_base)
        self.autoreset = autoreset
        self.aug_flags = (AUG_PERMUTE if "permute" in augment else 0) | (AUG_ROT90 if "rot90" in augment else 0)
        # the vector env's state only evolves through the kernels, so redundant zero-fills of `selected` can be elided
        self.flags = self.batch.elide_flag
        if autoreset is True:
            self.flags |= STEP_AUTORESET
        elif autoreset == "resample":
            self.flags |= STEP_RESAMPLE
        elif autoreset:
            raise ValueError("autoreset must be False, True or 'resample'")
        self.max_episode_steps = max_episode_steps
        if max_episode_steps is not None:
            self.batch.set_truncation(int(max_episode_steps))
            self.flags |= STEP_TRUNCATE
        self.dense_reward = bool(dense_reward)
        if self.dense_reward:
            self.batch.set_dense_output()
            self.flags |= STEP_DENSE
        self.adaptation = True
        self._no_trunc = torch.zeros(self.N, dtype=torch.bool, device=self.device)
        self._obs = self._build_obs()

    # ---- observation = live device state -------------------------------------------------------------
    def _build_obs(self):
        b = self.batch
        obs = {"trials_remain": b.field("trials_remain"), "terminated": b.field("terminated"),
               "input": b.plane("input"), "input_dim": b.field("input_dim"),
               "grid": b.plane("grid"), "grid_dim": b.field("grid_dim")}
        if "clip" in b.planes:
            obs["clip"] = b.plane("clip")
            obs["clip_dim"] = b.field("clip_dim")
        if "selected" in b.planes:
            obs["selected"] = b.plane("selected")
            obs["object_states"] = {
                "active": b.field("active"), "object": b.plane("object"), "object_sel": b.plane("object_sel"),
                "object_dim": b.field("object_dim"), "object_pos": b.field("object_pos"),
                "background": b.plane("background"), "rotation_parity": b.field("rotation_parity")}
        return obs

    def _info(self):
        """`info` of reset / step: zero-copy device views; the two entries that need a gather through the task table
        (`task_index`, `subprob_index`) are produced when first read — a step that nobody asks for them launches nothing extra."""
        b = self.batch
        if getattr(self, "_info_views", None) is None:  # the views themselves never change: built once
            self._info_views = {"input": b.plane("input"), "input_dim": b.field("input_dim"), "answer": b.plane("answer"),
                                "answer_dim": b.field("answer_dim"), "steps": b.cnt[:, 0], "submit_count": b.cnt[:, 1]}
        info = _LazyInfo(self._info_views)
        if hasattr(b, "cur_task"):  # device tensors: which task-table entry / problem / pair every env runs right now
            info["table_index"] = b.cur_task
            entry = lambda: b.cur_task.long().clamp_min(0)  # noqa: E731
            info.lazy("task_index", lambda: self._entry_problem[entry()])
            info.lazy("subprob_index", lambda: self._entry_sub[entry()])
        return info

    # ---- task table: Loader.parse's output, uploaded once -----------------------------------------------
    def _build_task_table(self):
        """Flattens Loader.data (loader.py:89-113) into one device table: all demo pairs, then all test pairs;
        per-task offsets/counts let `reset` turn (prob_index, subprob_index) into a table index."""
        data = self.loader.data
        ins, outs, eprob, esub = [], [], [], []
        self._off = {True: np.zeros(len(data), np.int64), False: np.zeros(len(data), np.int64)}
        self._cnt = {True: np.zeros(len(data), np.int64), False: np.zeros(len(data), np.int64)}
        for adaptation, (ii, oi) in ((True, (0, 1)), (False, (2, 3))):
            for t, task in enumerate(data):
                self._off[adaptation][t] = len(ins)
                self._cnt[adaptation][t] = len(task[ii])
                ins.extend(task[ii])
                outs.extend(task[oi])
                eprob.extend([t] * len(task[ii]))
                esub.extend(range(len(task[ii])))
        self.batch.set_task_table(ins, outs)
        self._entry_problem = torch.as_tensor(np.asarray(eprob, np.int64), device=self.device)
        self._entry_sub = torch.as_tensor(np.asarray(esub, np.int64), device=self.device)
        self._sampler_mode = None

    def _install_sampler(self, adaptation):
        """Candidates of the device-side draw: the problems that have at least one pair of the requested kind."""
        if self._sampler_mode == adaptation:
            return
        valid = np.nonzero(self._cnt[adaptation] > 0)[0]
        if valid.size == 0:
            raise ValueError("no task has a pair of the requested kind")
        self.sampler_problems = valid  # sampler index -> loader problem index
        self.batch.set_sampler(self._off[adaptation][valid], self._cnt[adaptation][valid], self.seed, self.env_base,
                               self.aug_flags)
        self._sampler_mode = adaptation

    def reset(self, seed=None, options=None, env_mask=None):
        """options as base.py:87-93 (prob_index / subprob_index may be ints or per-env sequences; adaptation;
        reset_on_submit).  Without prob_index / subprob_index the tasks are drawn on the device (keyed by the global env
        id and the env's episode count).  env_mask (bool [N], host or device) restricts the reset to some envs."""
        options = options or {}
        self.flags = (self.flags | STEP_RESET_ON_SUBMIT) if options.get("reset_on_submit") else (self.flags & ~STEP_RESET_ON_SUBMIT)
        adaptation = True if options.get("adaptation") is None else bool(options.get("adaptation"))
        self.adaptation = adaptation
        if self.loader is None:
            raise ValueError("ARCVecEnv needs a data_loader (or write tasks with batch.set_tasks and call batch.reset)")
        if not hasattr(self, "_off"):
            self._build_task_table()
        if seed is not None:
            self.seed = int(seed)
            self.rng = np.random.default_rng(seed)
            self._sampler_mode = None
            self.batch.__dict__.pop("episode", None)  # a new seed restarts the per-env draw streams
        self._install_sampler(adaptation)
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=self.device).to(torch.uint8)
        pidx, sidx = options.get("prob_index"), options.get("subprob_index")
        if pidx is None and sidx is None:
            self.batch.reset_sampled(mask)
            return self._obs, self._info()
        n_tasks = len(self.loader.data)
        p = self.rng.integers(0, n_tasks, self.N) if pidx is None else np.broadcast_to(np.asarray(pidx, np.int64), (self.N,))
        if ((p < 0) | (p >= n_tasks)).any():
            raise AssertionError(f"Problem indices should be in [0, {n_tasks}).")  # loader.py:55
        cnt = self._cnt[adaptation][p]
        if (cnt == 0).any():
            raise ValueError("a selected task has no pair of the requested kind")
        s_ = (self.rng.random(self.N) * cnt).astype(np.int64) if sidx is None else np.broadcast_to(np.asarray(sidx, np.int64), (self.N,))
        if ((s_ < 0) | (s_ >= cnt)).any():
            raise IndexError("subprob_index out of range")
        idx = torch.from_numpy((self._off[adaptation][p] + s_).astype(np.int32)).to(self.device)
        self.batch.reset_from_table(idx, mask)
        m = slice(None) if mask is None else mask.bool()
        self.batch.cur_task[m] = idx[m]
        return self._obs, self._info()

    # ---- step ------------------------------------------------------------------------------------------
    def _apply_host_ops(self, operation, action_of):
        """Table slots holding arbitrary Python callables (SURVEY.md §8b "custom ops"): the kernel counted the step, the
        callable now runs on the host on the fetched state of every env that chose such a slot.  Slow path."""
        op = operation.to("cpu").numpy()
        from .base import AbstractARCEnv
        for n in np.nonzero(np.isin(op, self._host_slots))[0]:
            state = AbstractARCEnv._state_from_device(self.batch, int(n))
            self.operations[int(op[n])](state, action_of(int(n)))
            AbstractARCEnv._state_to_device(self.batch, state, int(n))

    def _ret(self, reward, term, operation=None, action_of=None):
        b = self.batch
        if self._host_slots and operation is not None:
            self._apply_host_ops(operation, action_of)
            if self.flags & STEP_PACK_OBS:  # the step kernel packed its rows before the host callables ran: pack again
                b.packed_obs(b.packed)
        if self.dense_reward:
            d = b.dense.to(torch.float32)
            reward = reward.to(torch.float32) * 100.0 - 1.0 + d[:, 0] / d[:, 1]
        trunc = b.trunc.view(torch.bool) if self.max_episode_steps is not None else self._no_trunc  # (0/1 bytes: zero-copy)
        return self._obs, reward, term.view(torch.bool), trunc, self._info()

    def enable_packed_rows(self):
        """From now on every step also writes `batch.packed` ([N, R] uint8: grid | grid_dim | reward | terminated per env) from
        inside the step kernel (STEP_PACK_OBS) — what ShardedVecEnv.gather sends to a central learner."""
        packed = self.batch.set_packed_output()
        self.flags |= STEP_PACK_OBS
        return packed

    def step_bbox(self, bbox, operation):
        def action_of(n):  # BBoxWrapper.action (bbox.py:22-30) for one env, only needed by host-applied ops
            x1, y1, x2, y2 = (int(v) for v in bbox[n].tolist())
            sel = np.zeros((self.H, self.W), np.int8)
            sel[min(x1, x2):max(x1, x2) + 1, min(y1, y2):max(y1, y2) + 1] = 1
            return {"selection": sel, "operation": int(operation[n])}
        return self._ret(*self.batch.step_bbox(bbox, operation, self.flags), operation, action_of)

    def step_point(self, xy, operation):
        def action_of(n):
            sel = np.zeros((self.H, self.W), np.int8)
            sel[int(xy[n, 0]), int(xy[n, 1])] = 1
            return {"selection": sel, "operation": int(operation[n])}
        return self._ret(*self.batch.step_point(xy, operation, self.flags), operation, action_of)

    def step(self, action):
        sel, operation = action["selection"], action["operation"]
        return self._ret(*self.batch.step_mask(sel, operation, self.flags), operation,
                         lambda n: {"selection": sel[n].cpu().numpy(), "operation": int(operation[n])})

    def rollout_bbox(self, bbox, operation):
        """T steps in ONE launch: bbox int32 [T,N,4], operation int32 [T,N] -> (obs, reward [T,N], terminated [T,N]).
        For callers that already hold the action sequence (trace replay, scripted policies); the state is only
        observable after the last step."""
        reward, term = self.batch.rollout(bbox, operation, self._rollout_flags())
        return self._obs, reward, term.bool(), self._info()

    def rollout_point(self, xy, operation):
        reward, term = self.batch.rollout(xy, operation, self._rollout_flags(), point=True)
        return self._obs, reward, term.bool(), self._info()

    def _rollout_flags(self):
        if self._host_slots or self.flags & (STEP_RESAMPLE | STEP_TRUNCATE | STEP_DENSE):
            raise NotImplementedError("rollouts support plain and same-task autoreset envs with device-only op tables")
        return self.flags & ~STEP_PACK_OBS  # (only the final state of a rollout is observable: no per-step packed rows)

    def flat_obs(self, out=None, filtered=False):
        """The observation as one [N, L] int8 tensor in FlattenObservation key order (what the reference's policies
        consume, agents/models/GPTPolicy.py:17-35); filtered=True: the FilterO2ARC subset (agents/env.py:109-126)."""
        return self.batch.flat_obs(out, filtered)

    def check_errors(self):
        """Raises if any env saw an out-of-range op / out-of-domain Rotate / bad task index since the last check
        (the reference raises IndexError / ValueError at the offending step)."""
        st = self.batch.status()
        if st & ST_BAD_OP:
            raise IndexError("an env received an operation index outside its table")
        if st & ST_ROTATE_DOMAIN:
            raise ValueError("Rotate/Flip outside its domain (object.py:45 / int8 overflow of object_pos)")
        if st & ST_BAD_TASK:
            raise IndexError("a reset named a task-table index outside the table")
        if st & ST_BAD_SELECTION:
            raise IndexError("a point outside the grid plane / a negative selection coordinate (the reference's wrappers raise "
                             "IndexError or wrap the index, bbox.py:22-30,43-49)")

    def close(self):
        self.batch = None
