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

from .. import actions, sampling
from ..engine import (AUG_PERMUTE, AUG_ROT90, EnvBatch, STEP_AUTORESET, STEP_DENSE, STEP_FLAT_OBS, STEP_PACK_OBS, STEP_RESAMPLE,
                      STEP_RESET_ON_SUBMIT, STEP_ROWS_INCREMENTAL, STEP_TRUNCATE, ST_AUG_DOMAIN, ST_BAD_OP, ST_BAD_SELECTION, ST_BAD_TASK, ST_ROTATE_DOMAIN,
                      check_grid_size)


def _table_of(env_cls, **ctor_kw):
    """The class's operation table: `create_operations()` is the reference's plugin point (base.py:140-142), so an
    override of it is honoured.  Most overrides only build a list (agents/env.py:23-28), so the method is first called on a bare
    instance — no env, no device batch; an override that reads instance state set up by __init__ makes that probe raise, and the
    class is then constructed for real (same kwargs as the vector env) to ask it."""
    probe = object.__new__(env_cls)
    try:
        return list(probe.create_operations())
    except Exception:  # noqa: BLE001 - whatever the override needed from __init__
        return list(env_cls(**ctor_kw).operations)


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

    def rearm(self, thunks):
        """Makes the given entries pending again (their cached values are dropped): the same `info` object serves every step."""
        for k in thunks:
            dict.pop(self, k, None)
        self._thunks.update(thunks)

    def copy(self):
        self._force()
        return dict(self)

    def pop(self, key, *default):
        self._force(key)
        return dict.pop(self, key, *default)

    def setdefault(self, key, default=None):
        self._force(key)
        return dict.setdefault(self, key, default)


class CapturedSteps:
    """K consecutive steps of an ARCVecEnv captured into ONE hipGraph (ARCVecEnv.capture).  The action buffers are part of the
    object: write the next K actions into `.payload` / `.operation` (in place), then `replay()` — one host call per K steps, so the
    front-end runs at the kernel's rate instead of the interpreter's.
      payload      the selection payload the graph reads, [K, N, ...] in the captured ingress form
      operation    int32 [K, N] (None for the "bbox5" record form: the op is the record's fifth field)
      reward       [K, N] int32 — float32 with dense_reward — of the last replay
      terminated   [K, N] bool,  truncated [K, N] bool
    A captured graph keeps the tables / outputs it was captured with (every launch holds its parameters by value)."""

    def __init__(self, venv, graph, payload, operation, reward, term, trunc, keep=()):
        self.venv, self.graph, self.payload, self.operation = venv, graph, payload, operation
        self.reward, self.terminated, self.truncated = reward, term, trunc
        self.steps = int(payload.shape[0])
        self._keep = keep  # every buffer the captured launches write (raw rewards, dense pairs, ...) lives as long as the graph

    def replay(self):
        self.graph.replay()
        return self.venv._obs, self.reward, self.terminated, self.truncated


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
        augment: subset of ("permute", "rot90") (True = both) — task augmentation (agents/env.py:31-42) whenever a task is
                 LOADED: every `reset` (device-drawn or with prob_index / subprob_index) and every autoreset="resample" restart;
                 autoreset=True restarts the env on the planes it already holds, i.e. on the same, already augmented task.
        Everything `reset` / `step_*` return lives on the device and is a VIEW of this env's buffers (obs planes, reward,
        terminated, truncated, info entries): the next step overwrites them in place — copy what must outlive it."""
        self.env_cls, self.N = env_cls, int(num_envs)
        self.H, self.W = int(max_grid_size[0]), int(max_grid_size[1])
        check_grid_size(self.H, self.W)
        self.colors = colors
        if max_trial is None:
            max_trial = 3 if env_cls.KIND == "arc" else -1  # the classes' defaults (arcenv.py:79, o2arcenv.py:14)
        self.max_trial = max_trial
        self.loader = data_loader
        self.operations = list(operations) if operations is not None else _table_of(
            env_cls, data_loader=data_loader, max_grid_size=max_grid_size, colors=colors, max_trial=max_trial, device=device)
        self.op_names = ["".join(map(str.capitalize, op.__name__.split("_"))) for op in self.operations]
        self._host_slots = actions.host_slots(self.operations)
        self.batch = EnvBatch(self.N, self.H, self.W, max_trial, env_cls.KIND, device)
        self.batch.set_op_table(actions.table_descs(self.operations))
        self.device = self.batch.device
        self.rng = rng if rng is not None else np.random.default_rng(seed)
        self.seed = int(seed) if seed is not None else int(self.rng.integers(0, 2**63))
        self.env_base = int(env_base)
        self.autoreset = autoreset
        if augment is True:
            augment = ("permute", "rot90")
        elif not augment:
            augment = ()
        elif isinstance(augment, str):
            augment = (augment,)
        if set(augment) - {"permute", "rot90"}:
            raise ValueError('augment: a subset of ("permute", "rot90"), or True for both')
        self.aug_flags = (AUG_PERMUTE if "permute" in augment else 0) | (AUG_ROT90 if "rot90" in augment else 0)
        # the vector env's state only evolves through the kernels, so redundant zero-fills of `selected` can be elided — unless the
        # table holds host callables, which may write anything into the state they are handed
        self.flags = 0 if self._host_slots else self.batch.elide_flag
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
        """`info` of reset / step: zero-copy device views of live buffers (like `obs`, the next step updates them in place — ONE dict
        object serves every step); the two entries that need a gather through the task table (`task_index`, `subprob_index`) are
        produced when first read after a step — a step that nobody asks for them launches nothing extra."""
        b = self.batch
        info = getattr(self, "_info_obj", None)
        if info is None:  # the views themselves never change: built once
            info = self._info_obj = _LazyInfo({"input": b.plane("input"), "input_dim": b.field("input_dim"), "answer": b.plane("answer"),
                                               "answer_dim": b.field("answer_dim"), "steps": b.cnt[:, 0], "submit_count": b.cnt[:, 1]})
            self._info_thunks = None
        if self._info_thunks is None and hasattr(b, "cur_task"):  # which task-table entry / problem / pair every env runs right now
            info["table_index"] = b.cur_task
            entry = lambda: b.cur_task.long().clamp_min(0)  # noqa: E731
            self._info_thunks = {"task_index": lambda: self._entry_problem[entry()], "subprob_index": lambda: self._entry_sub[entry()]}
        if self._info_thunks is not None:
            info.rearm(self._info_thunks)
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
        # per table entry: does a quarter turn of the pair still fit the H x W plane? (non-square max_grid_size only)
        self._entry_turns = np.asarray([np.shape(a)[1] <= self.H and np.shape(a)[0] <= self.W and np.shape(b)[1] <= self.H and np.shape(b)[0] <= self.W
                                        for a, b in zip(ins, outs)], bool)
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
        mask = None if env_mask is None else torch.as_tensor(env_mask, device=self.device).to(torch.uint8)
        if seed is not None:
            self.seed = int(seed)
            self.rng = np.random.default_rng(seed)
            self._sampler_mode = None
            if hasattr(self.batch, "episode"):  # a new seed restarts the draw streams — of the envs being reset only
                if mask is None:
                    self.batch.episode.zero_()
                else:
                    self.batch.episode[mask.bool()] = 0
        self._install_sampler(adaptation)
        pidx, sidx = options.get("prob_index"), options.get("subprob_index")
        if pidx is None and sidx is None:
            self.batch.reset_sampled(mask)
            self._refresh_rows()
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
        m = slice(None) if mask is None else mask.bool()
        if self.aug_flags:
            # the caller chose the tasks; the augmentation is still drawn — what the device would draw for (seed, global env id,
            # episode), so a run is reproducible and independent of the sharding — and the envs' episode counters advance
            ep = self.batch.episode.cpu().numpy()
            k, perm = sampling.draw_aug_batch(self.seed, self.env_base + np.arange(self.N), ep, self.aug_flags)
            # the rule of the device-drawn path (arcle_wave.h load_task, soften): a quarter turn that does not fit a non-square plane is
            # dropped (k & 2) — so every env IS loaded and the episode / cur_task bookkeeping below is true for all of them
            k = np.where(self._entry_turns[idx.cpu().numpy()], k, k & 2).astype(k.dtype)
            self.batch.reset_from_table(idx, mask, k if self.aug_flags & AUG_ROT90 else None, perm if self.aug_flags & AUG_PERMUTE else None)
            self.batch.episode[m] += 1
        else:
            self.batch.reset_from_table(idx, mask)
        self.batch.cur_task[m] = idx[m]
        self._refresh_rows()
        return self._obs, self._info()

    # ---- step ------------------------------------------------------------------------------------------
    def _ended(self):
        """bool [N] (device): envs whose episode is over (terminated, or out of steps) — the ones the NEXT step auto-resets."""
        b = self.batch
        e = b.rec[:, 11] != 0
        if self.max_episode_steps is not None:
            e = e | (b.cnt[:, 0] >= int(self.max_episode_steps))
        return e

    def _apply_host_ops(self, operation, action_of, skip, reward, term):
        """Table slots holding arbitrary Python callables (SURVEY.md §8b "custom ops"): the kernel counted the step, the
        callable now runs on the host on the fetched state of every env that chose such a slot — except the envs the kernel
        auto-reset in this step (`skip`: their action was not executed).  `terminated` and the reward of a host-applied LAST slot
        are re-evaluated on the state the callable left (o2arcenv.py:121-128), as the single-env class does.  Slow path."""
        op = operation.to("cpu").numpy()
        skip = None if skip is None else skip.cpu().numpy()
        from .base import AbstractARCEnv
        b = self.batch
        for n in np.nonzero(np.isin(op, self._host_slots))[0]:
            if skip is not None and skip[n]:
                continue
            state = AbstractARCEnv._state_from_device(b, int(n))
            self.operations[int(op[n])](state, action_of(int(n)))
            AbstractARCEnv._state_to_device(b, state, int(n))
            term[n] = int(state["terminated"][0] != 0)
            if int(op[n]) == len(self.operations) - 1:
                ah, aw = (int(v) for v in b.field("answer_dim")[n].tolist())
                same = tuple(int(v) for v in state["grid_dim"]) == (ah, aw) and np.array_equal(
                    state["grid"][:ah, :aw], b.plane("answer")[n, :ah, :aw].cpu().numpy())
                reward[n] = int(same)

    def _ret(self, reward, term, operation=None, action_of=None, skip=None):
        b = self.batch
        if self._host_slots and operation is not None:
            self._apply_host_ops(operation, action_of, skip, reward, term)
            self._refresh_rows()
            if self.flags & STEP_PACK_OBS:  # the step kernel packed its rows before the host callables ran: pack again
                b.packed_obs(b.packed)
        if self.dense_reward:
            reward = self._dense(reward, b.dense)
        trunc = b.trunc.view(torch.bool) if self.max_episode_steps is not None else self._no_trunc  # (0/1 bytes: zero-copy)
        return self._obs, reward, term.view(torch.bool), trunc, self._info()

    @staticmethod
    def _dense(reward, dense):
        """sparse * 100 - 1 + correct / total (agents/env.py:44-58) from the kernel's integer pairs; a step that executed no action
        (the auto-reset step of an env, a skipped step) carries the pair (0, 0) and gets reward 0."""
        d = dense.to(torch.float32)
        tot = d[..., 1]
        r = reward.to(torch.float32) * 100.0 - 1.0 + d[..., 0] / tot.clamp_min(1.0)
        return torch.where(tot > 0, r, torch.zeros_like(r))

    def _host_skip(self):
        """With host-applied slots and an auto-resetting env: which envs will be re-initialised (not stepped) by the next launch."""
        if self._host_slots and self.flags & (STEP_AUTORESET | STEP_RESAMPLE):
            return self._ended().clone()
        return None

    def enable_flat_rows(self, filtered=True):
        """From now on every step also keeps `rows` — int8 [N, L], the FlattenObservation row of every env (filtered=True: the
        FilterO2ARC subset the reference's policies consume, agents/env.py:109-126) — up to date from inside the step kernel.  The
        buffer is a live mirror like `obs`: a step rewrites, per env, only the scalars and the segments of the planes that step
        changed (ARCLE_STEP_ROWS_INCREMENTAL); resets and state ingests rewrite it in full.  Copy it to keep a step's rows."""
        self.rows = self.batch.set_flat_output(filtered)
        self._rows_filtered = bool(filtered)
        self.flags |= STEP_FLAT_OBS | STEP_ROWS_INCREMENTAL
        self._refresh_rows()
        return self.rows

    def _refresh_rows(self):
        if self.flags & STEP_FLAT_OBS:
            self.batch.flat_obs(out=self.batch._flat_buf, filtered=self._rows_filtered)

    def enable_packed_rows(self, out=None):
        """From now on every step also writes `batch.packed` ([N, R] uint8: grid | grid_dim | reward | terminated per env) from
        inside the step kernel (STEP_PACK_OBS) — what ShardedVecEnv.gather sends to a central learner.  `out`: caller-owned buffer."""
        packed = self.batch.set_packed_output(out)
        self.flags |= STEP_PACK_OBS
        return packed

    def step_bbox(self, bbox, operation, next_operation=None):
        """bbox int32 [N, 4], operation int32 [N] (device).  next_operation: round 4's hint of the FOLLOWING step's operations — launches
        order themselves since round 5 (the object operations go to the waves that start first, derived from THIS step's operations), so
        it is accepted and ignored."""

        def action_of(n):  # BBoxWrapper.action (bbox.py:22-30) for one env, only needed by host-applied ops
            x1, y1, x2, y2 = (int(v) for v in bbox[n].tolist())
            sel = np.zeros((self.H, self.W), np.int8)
            sel[min(x1, x2):max(x1, x2) + 1, min(y1, y2):max(y1, y2) + 1] = 1
            return {"selection": sel, "operation": int(operation[n])}
        skip = self._host_skip()
        return self._ret(*self.batch.step_bbox(bbox, operation, self.flags), operation, action_of, skip)

    def step_point(self, xy, operation, next_operation=None):  # (next_operation: accepted and ignored, see step_bbox)
        def action_of(n):
            sel = np.zeros((self.H, self.W), np.int8)
            sel[int(xy[n, 0]), int(xy[n, 1])] = 1
            return {"selection": sel, "operation": int(operation[n])}
        skip = self._host_skip()
        return self._ret(*self.batch.step_point(xy, operation, self.flags), operation, action_of, skip)

    def step(self, action):
        sel, operation = action["selection"], action["operation"]
        skip = self._host_skip()
        return self._ret(*self.batch.step_mask(sel, operation, self.flags), operation,
                         lambda n: {"selection": sel[n].cpu().numpy(), "operation": int(operation[n])}, skip)

    def step_bbox5(self, act5, next_act5=None):
        """The BBoxWrapper action as it is sampled (examples/example_bbox.py:13-15): int32 [N, 5] = (x1, y1, x2, y2, operation), ONE
        array (device, or pinned host memory — the kernel then reads it across PCIe, no copy in front of the step).  next_act5: accepted
        and ignored (see step_bbox's next_operation)."""
        if self._host_slots:
            return self.step_bbox(act5[:, :4].contiguous(), act5[:, 4].contiguous())
        return self._ret(*self.batch.step_bbox5(act5, self.flags))

    def step_bits(self, bits, operation):
        """Boolean selection masks, bit-packed: uint8 [N, 128] (`pack_masks` converts [N,H,W] masks), operation int32 [N]."""
        if self._host_slots:
            raise NotImplementedError("host-applied table slots need the int8 mask form (step)")
        return self._ret(*self.batch.step_bits(bits, operation, self.flags))

    def pack_masks(self, sel, out=None):
        return self.batch.pack_mask_bits(sel, out)

    # ---- K steps per host call ---------------------------------------------------------------------------
    def _many_ok(self):
        if self._host_slots:
            raise NotImplementedError("multi-step calls need a device-only op table (no host callables)")

    def _redirect(self, i, trunc, dense):
        b = self.batch
        if trunc is not None:
            b.L.arcle_set_truncation(b._h, trunc[i].data_ptr(), int(self.max_episode_steps))
        if dense is not None:
            b.L.arcle_set_dense_output(b._h, dense[i].data_ptr())

    def _restore_outputs(self):
        b = self.batch
        if self.max_episode_steps is not None:
            b.L.arcle_set_truncation(b._h, b.trunc.data_ptr(), int(self.max_episode_steps))
        if self.dense_reward:
            b.L.arcle_set_dense_output(b._h, b.dense.data_ptr())

    def _enqueue_steps(self, form, payload, operation, reward, term, trunc, dense):
        """K step launches on the current stream, step i writing reward[i] / term[i] / trunc[i] / dense[i]."""
        b = self.batch
        K = int(payload.shape[0])
        if trunc is None and dense is None:  # nothing to redirect per step: ONE call into the library enqueues all K launches
            b.step_many(form, payload, operation, self.flags, reward, term)
            return
        fn = {"mask": b.L.arcle_step_mask, "bbox": b.L.arcle_step_bbox, "point": b.L.arcle_step_point, "bits": b.L.arcle_step_bits}.get(form)
        st = b._stream()
        for i in range(K):
            self._redirect(i, trunc, dense)
            if form == "bbox5":
                rc = b.L.arcle_step_bbox5(b._h, payload[i].data_ptr(), reward[i].data_ptr(), term[i].data_ptr(), self.flags, st)
            else:
                rc = fn(b._h, payload[i].data_ptr(), operation[i].data_ptr(), reward[i].data_ptr(), term[i].data_ptr(), self.flags, st)
            b._check(rc, "arcle_step")
        self._restore_outputs()

    def _many_buffers(self, K):
        dev, N = self.device, self.N
        reward = torch.zeros((K, N), dtype=torch.int32, device=dev)
        term = torch.zeros((K, N), dtype=torch.uint8, device=dev)
        trunc = torch.zeros((K, N), dtype=torch.uint8, device=dev) if self.max_episode_steps is not None else None
        dense = torch.zeros((K, N, 2), dtype=torch.int32, device=dev) if self.dense_reward else None
        return reward, term, trunc, dense

    def _check_many(self, form, payload, operation):
        shapes = {"bbox": (self.N, 4), "point": (self.N, 2), "bbox5": (self.N, 5), "mask": (self.N, self.H, self.W), "bits": (self.N, 128)}
        dt = {"mask": torch.int8, "bits": torch.uint8}.get(form, torch.int32)
        assert form in shapes, f"unknown action form {form!r}"
        on_dev = payload.device == self.device or (form == "bbox5" and payload.is_pinned())  # (records of a host-resident policy: pinned memory)
        assert payload.dtype == dt and tuple(payload.shape[1:]) == shapes[form] and payload.is_contiguous() and on_dev
        if form != "bbox5":
            assert operation is not None and operation.dtype == torch.int32 and tuple(operation.shape) == (payload.shape[0], self.N)
            assert operation.is_contiguous() and operation.device == self.device

    def step_many(self, payload, operation=None, form="bbox"):
        """K consecutive steps enqueued by ONE call: payload [K, N, ...] in the action form `form` ("bbox" int32 [K,N,4] | "point"
        [K,N,2] | "bbox5" [K,N,5] | "mask" int8 [K,N,H,W] | "bits" uint8 [K,N,128]), operation int32 [K, N] (None for "bbox5").
        Returns (obs, reward [K,N], terminated [K,N], truncated [K,N], info): obs / info are the live state after the LAST step,
        the other three hold every step.  Same semantics as K step_* calls (for callers that hold the next K actions: scripted
        policies, action chunks, trace replay with observable intermediate rewards)."""
        self._many_ok()
        self._check_many(form, payload, operation)
        reward, term, trunc, dense = self._many_buffers(int(payload.shape[0]))
        self._enqueue_steps(form, payload, operation, reward, term, trunc, dense)
        r = self._dense(reward, dense) if dense is not None else reward
        tr = trunc.view(torch.bool) if trunc is not None else torch.zeros_like(term, dtype=torch.bool)
        return self._obs, r, term.view(torch.bool), tr, self._info()

    def capture(self, payload, operation=None, form="bbox"):
        """Captures K = payload.shape[0] consecutive steps — reading their actions from `payload` / `operation` as they are at
        REPLAY time — into one hipGraph and returns a CapturedSteps.  The usual loop becomes
            cs = venv.capture(bbox_buf, op_buf)               # once
            while training: policy writes K actions into cs.payload / cs.operation;  obs, r, term, trunc = cs.replay()
        i.e. one host call per K steps (a 5 us kernel cannot be fed step by step from Python)."""
        self._many_ok()
        self._check_many(form, payload, operation)
        K = int(payload.shape[0])
        reward, term, trunc, dense = self._many_buffers(K)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            self._enqueue_steps(form, payload, operation, reward, term, trunc, dense)
            r = self._dense(reward, dense) if dense is not None else reward
        tr = trunc.view(torch.bool) if trunc is not None else torch.zeros_like(term, dtype=torch.bool)
        return CapturedSteps(self, g, payload, operation, r, term.view(torch.bool), tr, keep=(reward, term, trunc, dense))

    # ---- state in / out ------------------------------------------------------------------------------------
    def state_rows(self, out=None):
        """The state dict of every env as one row: int8 [N, L] in FlattenObservation order (6314 bytes for 30 x 30)."""
        return self.batch.get_state_rows(out)

    def set_state_rows(self, rows, env_mask=None):
        """Overwrites the state of the (masked) envs from rows as `state_rows` returns them."""
        self.batch.set_state_rows(rows, env_mask)
        self.flags &= ~self.batch.elide_flag  # states from outside may break the invariant the zero-fill elision rests on
        self._refresh_rows()

    def transition(self, rows, action, src_env=None, out=None, in_place=False):
        """The reference's `transition(state, action)` (o2arcenv.py:149-151; README: `env.transition(deepcopy(state), action)`) for
        a batch: rows int8 [M, L] = M states (as `state_rows` / a previous `transition` returns them — M is NOT tied to num_envs),
        action = {"selection": [M,H,W] mask | "bbox": int32 [M,4] | "point": int32 [M,2], "operation": int32 [M]}; src_env int32
        [M] = the env whose task (answer) row m belongs to (default: env m).  Returns (rows_out [M, L], reward int32 [M], terminated
        bool [M]).  Nothing of this env's own state is touched: expanding thousands of hypothetical states per launch is the point.
        in_place=True: `rows` (a tensor an earlier `transition` returned, or any view of a [M, 16-byte-multiple] buffer) is overwritten
        with the successor states — the kernel then rewrites only the planes the op changed, about half the time of the
        out-of-place form (walking M trajectories forward rather than branching)."""
        b = self.batch
        if in_place:
            stride = rows.stride(0)
            if rows.dim() != 2 or rows.stride(1) != 1 or stride % 16 or stride < ((b.state_row_size() + 15) & ~15) or rows.data_ptr() % 16:
                raise ValueError("in_place: rows must be a view of a 16-byte aligned [M, stride] int8 buffer, stride a multiple of 16 >= the row length")
            out = torch.as_strided(rows, (rows.shape[0], stride), (stride, 1))
        if "bbox" in action:
            form, pay = "bbox", action["bbox"].to(device=self.device, dtype=torch.int32).contiguous()
        elif "point" in action:
            form, pay = "point", action["point"].to(device=self.device, dtype=torch.int32).contiguous()
        else:
            form, pay = "mask", action["selection"].to(device=self.device, dtype=torch.int8).contiguous()
        op = action["operation"].to(device=self.device, dtype=torch.int32).contiguous()
        if src_env is not None:
            src_env = src_env.to(device=self.device, dtype=torch.int32).contiguous()
        fl = STEP_RESET_ON_SUBMIT if self.flags & STEP_RESET_ON_SUBMIT else 0
        rows_out, reward, term = b.transition_rows(rows, form, pay, op, src_env, out, flags=fl)
        return rows_out[:, :b.state_row_size()], reward, term.view(torch.bool)

    def autotune(self, payload, operation=None, form="bbox"):
        """Times every launch plan the library has for this env's steps (self-ordering or not, the cache policies of the speculative grid
        request, 4- or 8-wave workgroups) on K consecutive action batches of the caller — payload [K, N, ...], operation [K, N], device
        tensors: a representative stretch of the policy's output — and keeps the fastest for the steps that follow (arcle_autotune; the
        env state is saved and restored, the stream synchronised).  Plain / same-task-autoreset envs; returns the candidates, fastest first."""
        if self.flags & ~(STEP_AUTORESET | self.batch.elide_flag | STEP_PACK_OBS):
            raise NotImplementedError("autotune supports plain and same-task autoreset envs (other flag sets keep per-env side state it does not save)")
        return self.batch.autotune(form, payload, operation, self.flags)

    def get_state(self):
        """Checkpoint (cloned device tensors) of the whole batch: states, tasks, counters, task-draw positions."""
        return self.batch.get_state()

    def set_state(self, st):
        self.batch.set_state(st)
        self._refresh_rows()

    def rollout_bbox(self, bbox, operation, packed=None):
        """T steps in ONE launch: bbox int32 [T,N,4], operation int32 [T,N] -> (obs, reward [T,N], terminated [T,N]).
        For callers that already hold the action sequence (action chunks, trace replay, scripted policies).  The state dict is only
        observable after the last step; packed = a uint8 [T, N, batch.packed_obs_size()] device tensor additionally receives the packed
        observation row (grid | grid_dim | reward | terminated, `EnvBatch.unpack_obs`) of EVERY step."""
        reward, term = self.batch.rollout(bbox, operation, self._rollout_flags(), packed=packed)
        self._refresh_rows()  # (a rollout keeps no per-step flat rows: the live mirror is rewritten from the final state)
        return self._obs, reward, term.bool(), self._info()

    def rollout_point(self, xy, operation):
        reward, term = self.batch.rollout(xy, operation, self._rollout_flags(), point=True)
        self._refresh_rows()
        return self._obs, reward, term.bool(), self._info()

    def _rollout_flags(self):
        if self._host_slots or self.flags & (STEP_RESAMPLE | STEP_TRUNCATE | STEP_DENSE):
            raise NotImplementedError("rollouts support plain and same-task autoreset envs with device-only op tables")
        # (only the final state of a rollout is observable: no per-step packed / flat rows — the callers refresh the live rows afterwards)
        return self.flags & ~(STEP_PACK_OBS | STEP_FLAT_OBS | STEP_ROWS_INCREMENTAL)

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
            raise IndexError("a reset named a task-table index outside the table / a transition row named an env that does not exist")
        if st & ST_AUG_DOMAIN:
            raise ValueError("an explicit rot90 augmentation by an odd count does not fit a non-square max_grid_size: env left untouched")
        if st & ST_BAD_SELECTION:
            raise IndexError("a point outside the grid plane / a negative selection coordinate (the reference's wrappers raise "
                             "IndexError or wrap the index, bbox.py:22-30,43-49)")

    def close(self):
        self.batch = None
