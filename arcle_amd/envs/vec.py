"""ARCVecEnv — N independent ARCLE envs stepped by ONE kernel launch per step (the hot path the
throughput numbers are quoted on).  Observations are zero-copy torch views of the device state (like the
reference, `obs` IS the live state — o2arcenv.py:147), actions are device tensors:

    venv = ARCVecEnv(O2ARCv2Env, num_envs=8192, data_loader=loader, max_grid_size=(30, 30))
    obs, info = venv.reset()
    obs, reward, terminated, truncated, info = venv.step_bbox(bbox_i32[N,4], op_i32[N])     # BBoxWrapper form
    obs, reward, terminated, truncated, info = venv.step_point(xy_i32[N,2], op_i32[N])      # PointWrapper form
    obs, reward, terminated, truncated, info = venv.step({"selection": m[N,H,W], "operation": op[N]})
"""
import numpy as np
import torch

from .. import actions
from ..engine import EnvBatch, STEP_AUTORESET


class ARCVecEnv:
    def __init__(self, env_cls, num_envs, data_loader=None, max_grid_size=(30, 30), colors=10, max_trial=None,
                 device=None, autoreset=False, operations=None, rng=None):
        """env_cls: RawARCEnv / ARCEnv / O2ARCv2Env or a subclass (its `create_operations` / `default_operations`
        and KIND define the op table and the state planes).  `operations` overrides the table.
        autoreset=True gives Gymnasium next-step autoreset semantics on device (ARCLE_STEP_AUTORESET)."""
        self.env_cls, self.N = env_cls, int(num_envs)
        self.H, self.W = int(max_grid_size[0]), int(max_grid_size[1])
        self.colors = colors
        if max_trial is None:
            max_trial = 3 if env_cls.KIND == "arc" else -1  # the classes' defaults (arcenv.py:79, o2arcenv.py:14)
        self.max_trial = max_trial
        self.loader = data_loader
        self.operations = list(operations) if operations is not None else env_cls.default_operations()
        self.op_names = ["".join(map(str.capitalize, op.__name__.split("_"))) for op in self.operations]
        self.batch = EnvBatch(self.N, self.H, self.W, max_trial, env_cls.KIND, device)
        self.batch.set_op_table(actions.table_descs(self.operations))
        self.device = self.batch.device
        self.rng = rng if rng is not None else np.random.default_rng()
        # autoreset: False | True (Gymnasium next-step autoreset of the SAME task, inside the step kernel)
        #            | "resample" (same-step autoreset with a NEW random task from the device task table)
        self.autoreset = autoreset
        # the vector env's state only evolves through the kernels, so redundant zero-fills of `selected` can be elided
        self.flags = (STEP_AUTORESET if autoreset is True else 0) | self.batch.elide_flag
        self._gen = torch.Generator(device=self.batch.device)
        self._gen.manual_seed(int(self.rng.integers(0, 2**31)))
        self.task_index = np.zeros(self.N, np.int64)
        self.subprob_index = np.zeros(self.N, np.int64)
        self._truncated = torch.zeros(self.N, dtype=torch.bool, device=self.device)
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
        b = self.batch
        return {"input": b.plane("input"), "input_dim": b.field("input_dim"), "answer": b.plane("answer"),
                "answer_dim": b.field("answer_dim"), "steps": b.cnt[:, 0], "submit_count": b.cnt[:, 1],
                # host copies of the last explicit reset()'s choice; `table_index` (device, entry of the task table)
                # also follows autoreset="resample"
                "task_index": self.task_index, "subprob_index": self.subprob_index,
                "table_index": getattr(self, "table_index", None)}

    # ---- reset: task choice vectorised on the host (no per-env Python), grids come from the device task table ----
    def _build_task_table(self):
        """Flattens Loader.data (loader.py:89-113) into one device table: all demo pairs, then all test pairs;
        per-task offsets/counts let `reset` turn (prob_index, subprob_index) into a table index."""
        data = self.loader.data
        ins, outs = [], []
        self._off = {True: np.zeros(len(data), np.int64), False: np.zeros(len(data), np.int64)}
        self._cnt = {True: np.zeros(len(data), np.int64), False: np.zeros(len(data), np.int64)}
        for adaptation, (ii, oi) in ((True, (0, 1)), (False, (2, 3))):
            for t, task in enumerate(data):
                self._off[adaptation][t] = len(ins)
                self._cnt[adaptation][t] = len(task[ii])
                ins.extend(task[ii])
                outs.extend(task[oi])
        self.batch.set_task_table(ins, outs)
        self._dev_off = {k: torch.from_numpy(v).to(self.device) for k, v in self._off.items()}
        self._dev_cnt = {k: torch.from_numpy(v).to(self.device) for k, v in self._cnt.items()}

    def reset(self, seed=None, options=None, env_mask=None):
        """options as base.py:87-93 (prob_index / subprob_index may be ints or per-env sequences; adaptation).
        env_mask (bool [N], host or device) restricts the reset to some envs (they get NEW tasks)."""
        if seed is not None:
            self.rng = np.random.default_rng(seed)
        options = options or {}
        if options.get("reset_on_submit"):
            raise NotImplementedError("reset_on_submit=True is not supported on device (SURVEY.md A.6-7)")
        adaptation = True if options.get("adaptation") is None else bool(options.get("adaptation"))
        self.adaptation = adaptation
        if self.loader is None:
            raise ValueError("ARCVecEnv needs a data_loader (or write tasks with batch.set_tasks and call batch.reset)")
        if not hasattr(self, "_off"):
            self._build_task_table()
        n_tasks = len(self.loader.data)
        pidx, sidx = options.get("prob_index"), options.get("subprob_index")
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
        mask = None
        if env_mask is not None:
            mask = torch.as_tensor(env_mask, device=self.device).to(torch.uint8)
            keep = ~(mask.bool().cpu().numpy())
            p = np.where(keep, self.task_index, p)
            s_ = np.where(keep, self.subprob_index, s_)
        self.task_index, self.subprob_index = np.asarray(p).copy(), np.asarray(s_).copy()
        prev = getattr(self, "table_index", None)
        self.table_index = idx if (mask is None or prev is None) else torch.where(mask.bool(), idx, prev)
        self.batch.reset_from_table(idx, mask)
        return self._obs, self._info()

    def _resample_terminated(self, term):
        """autoreset='resample': envs that just terminated get a NEW random task, entirely on device."""
        ad = getattr(self, "adaptation", True)
        n_tasks = len(self.loader.data)
        p = torch.randint(0, n_tasks, (self.N,), device=self.device, generator=self._gen)
        cnt = self._dev_cnt[ad][p]
        s_ = (torch.rand(self.N, device=self.device, generator=self._gen) * cnt).long()
        s_ = torch.minimum(s_, cnt - 1)
        idx = (self._dev_off[ad][p] + s_).int()
        self.table_index = torch.where(term.bool(), idx, self.table_index)
        self.batch.reset_from_table(idx, term)

    # ---- step ------------------------------------------------------------------------------------------
    def _ret(self, reward, term):
        if self.autoreset == "resample":
            term = term.clone()  # the step outputs are overwritten by the next launch
            self._resample_terminated(term)
        return self._obs, reward, term.bool(), self._truncated, self._info()

    def step_bbox(self, bbox, operation):
        return self._ret(*self.batch.step_bbox(bbox, operation, self.flags))

    def step_point(self, xy, operation):
        return self._ret(*self.batch.step_point(xy, operation, self.flags))

    def step(self, action):
        return self._ret(*self.batch.step_mask(action["selection"], action["operation"], self.flags))

    def rollout_bbox(self, bbox, operation):
        """T steps in ONE launch: bbox int32 [T,N,4], operation int32 [T,N] -> (obs, reward [T,N], terminated [T,N]).
        For callers that already hold the action sequence (trace replay, scripted policies); the state is only
        observable after the last step."""
        reward, term = self.batch.rollout(bbox, operation, self.flags)
        return self._obs, reward, term.bool(), self._info()

    def rollout_point(self, xy, operation):
        reward, term = self.batch.rollout(xy, operation, self.flags, point=True)
        return self._obs, reward, term.bool(), self._info()

    def flat_obs(self, out=None):
        """The observation as one [N, L] int8 tensor in FlattenObservation key order (what the reference's policies
        consume, agents/models/GPTPolicy.py:17-35)."""
        return self.batch.flat_obs(out)

    def check_errors(self):
        """Raises if any env saw an out-of-range op / out-of-domain Rotate since the last check
        (the reference raises IndexError / ValueError at the offending step)."""
        st = self.batch.status()
        if st & 1:
            raise IndexError("an env received an operation index outside its table")
        if st & 2:
            raise ValueError("Rotate/Flip outside its domain (object.py:45 / int8 overflow of object_pos)")

    def close(self):
        self.batch = None
