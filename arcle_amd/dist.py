"""Multi-GPU: the env batch is partitioned across ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  Envs are independent — `step` has no cross-env dependency
(/root/reference/arcle/envs/o2arcenv.py:130-151 touches only `self`) — so the data path needs NO collective:
rank g owns the contiguous global env ids [lo_g, hi_g) (`shard_range`; any global size, the remainder goes to the lowest ranks)
and keys its device-side task draws by the GLOBAL env id (arcle_amd/sampling.py), so the trajectories do not depend on the
number of GPUs.  The only exchange that ever happens is the optional gather of what a central learner consumes,
(grid, grid_dim, reward, done): the step kernel itself packs its outputs into one 912-byte record per env (STEP_PACK_OBS, a fused
epilogue; arcle_pack_obs is the stand-alone form) and the records move with ONE all_gather_into_tensor (one-shot: every xGMI link
carries one shard).

The collective is 8-10x longer than a step (7.5 MB per GPU over 7 links ~ 50 us against a 5.5 us kernel), so it must not sit on
the step stream.  Three ways to keep it off the critical path, all provided here:
  * `gather_async()`  the all-gather runs on a SIDE stream behind an event; the step stream goes on.  Packed rows and gathered
                      tensors are double-buffered, so step t+1 may overwrite nothing the collective of step t still reads; the step
                      that re-enters a slot two windows later first waits for the collective that read it (whether or not the
                      caller ever called wait()), so deferring wait() by any number of steps cannot corrupt rows in flight.
  * `groups=2`        the shard is split into ping-pong groups with their own state: while group A's rows travel (and the learner
                      computes A's next actions) group B steps — the strict loop  a(t+1) = policy(obs(t))  leaves nothing else to
                      overlap with.
  * `every=K`         learners that consume K steps at a time: rows of K steps accumulate in one [K, n, R] buffer (the step kernel
                      writes slot t % K directly) and move with one K-times-larger collective — link latency paid once per K steps.
`capture()` records step + collective of K steps into one hipGraph (RCCL collectives are capturable); replay = one host call.

Every call works on torch's CURRENT stream, so ping-pong groups can also be given a HIP stream each (`with torch.cuda.stream(s[g]):
env.step_bbox(..., group=g)`): two free-running groups on two queues overlap one group's launch turnaround and tail with the other's ramp —
16 384 envs as 2 x 8192: 7.2 instead of 9.0-9.3 us per step of all envs (profiles/round5_experiments.txt §11); groups of 4096 or fewer lose.
"""
import torch
import torch.distributed as dist

from .engine import EnvBatch


def shard_range(global_envs, world_size, rank):
    """Contiguous global env ids owned by `rank`; the remainder goes to the lowest ranks."""
    base, rem = divmod(int(global_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def split_range(lo, hi, parts):
    """[lo, hi) cut into `parts` contiguous sub-ranges (the ping-pong groups of one shard)."""
    return [tuple(lo + v for v in shard_range(hi - lo, parts, g)) for g in range(parts)]


class GatherWork:
    """Handle of one gather in flight.  `wait()` orders the caller's current stream (CPU: the calling thread) behind the collective
    and returns (grid [G,H,W] int8, grid_dim [G,2] int8, reward [G] int32, terminated [G] bool) in global env order — views of a
    buffer that stays valid until the gather after the next one of the same group is issued (double buffering)."""

    def __init__(self, owner, group, slot, event=None, work=None):
        self.owner, self.group, self.slot, self.event, self.work = owner, group, slot, event, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
        if self.event is not None:
            torch.cuda.current_stream(self.owner.device).wait_event(self.event)
        return self.owner._unpack(self.group, self.slot)


class _Group:
    """One ping-pong group of a shard: its local vector env, global id range and the double-buffered rows."""

    def __init__(self, env, lo, hi):
        self.env, self.lo, self.hi, self.n = env, lo, hi, hi - lo
        self.packed = self.full = None  # [2][K] buffers, allocated on first use
        self.slot = 0                   # buffer pair the NEXT step writes
        self.t = 0                      # steps taken since the last every-K gather
        self.last = None                # (slot, steps) of the rows the last step(s) produced
        self.consumed = [None, None]    # per slot: event after which the consumer is done with full[slot]
        self.inflight = [None, None]    # per slot: (event | gloo work) of the gather that is still READING packed[slot]


class ShardedVecEnv:
    """This rank's shard of a global batch of `global_envs` envs.  `local_env_factory(n_local, lo, hi)` builds a local vector env
    for the global ids [lo, hi) — for the HIP path: `lambda n, lo, hi: ARCVecEnv(cls, n, loader, seed=S, env_base=lo, ...)`
    (env_base makes the device-side task draws follow the global env id).  groups > 1: the factory is called once per ping-pong
    group; step / gather calls then take `group=`.  every = K: rows of K steps travel together (see the module docstring)."""

    def __init__(self, global_envs, local_env_factory, group=None, fused_pack=True, groups=1, every=1, force_collective=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.global_envs = int(global_envs)
        self.lo, self.hi = shard_range(global_envs, self.world, self.rank)
        self.N = self.hi - self.lo
        self.every = int(every)
        self.n_groups = int(groups)
        # all_gather_into_tensor wants equal shards: every rank sends n_max rows per group (the ranks that own one env less pad)
        spans = [[split_range(*shard_range(global_envs, self.world, r), self.n_groups)[g] for r in range(self.world)] for g in range(self.n_groups)]
        self._n_max = [max(hi - lo for lo, hi in spans[g]) for g in range(self.n_groups)]
        self._spans = spans
        self.groups = [_Group(local_env_factory(hi - lo, lo, hi), lo, hi) for lo, hi in split_range(self.lo, self.hi, self.n_groups)]
        self.local = self.groups[0].env
        b0 = self.local.batch
        self.device = b0.device
        self._cuda = self.device.type == "cuda"
        self._side = torch.cuda.Stream(self.device) if self._cuda else None
        self._collective = self.world > 1 or force_collective
        # fused_pack: the local env's step kernel writes the packed rows itself (ARCVecEnv.enable_packed_rows), so that
        # gather() is the collective alone; otherwise gather() launches arcle_pack_obs first
        self.fused = bool(fused_pack) and hasattr(self.local, "enable_packed_rows")
        R = b0.packed_obs_size()
        for g, grp in enumerate(self.groups):
            nm = self._n_max[g]
            grp.packed = [torch.zeros((self.every, nm, R), dtype=torch.uint8, device=self.device) for _ in range(2)]
            # (gathered as the concatenation of the ranks' [K, n_max, R] blocks along dim 0: the output shape both RCCL and gloo take)
            grp.full = [torch.zeros((self.world * self.every, nm, R), dtype=torch.uint8, device=self.device) if self._collective else None
                        for _ in range(2)]
            if self.fused:
                grp.env.enable_packed_rows(grp.packed[0][0][:grp.n])
        # global order of a group's gathered rows: rank-major; index of the rows that exist (drops the padding of short shards)
        self._ids = []
        for g in range(self.n_groups):
            ids = torch.cat([torch.arange(lo, hi) for lo, hi in spans[g]])
            keep = torch.cat([r * self._n_max[g] + torch.arange(hi - lo) for r, (lo, hi) in enumerate(spans[g])])
            self._ids.append((ids.to(self.device), None if len(keep) == self.world * self._n_max[g] else keep.to(self.device)))

    # ---- local stepping: no communication ----------------------------------------------------------------------
    def reset(self, group=None, **kw):
        if group is None:
            out = [g.env.reset(**kw) for g in self.groups]
            return out[0] if self.n_groups == 1 else out
        return self.groups[group].env.reset(**kw)

    def _before_step(self, grp):
        if grp.t == 0:
            # a window starts in this slot: the gather that last read packed[slot] (issued two windows ago, maybe never waited for by
            # the caller — the collective is 8-10x longer than a step) must have finished before a step kernel / arcle_pack_obs
            # rewrites the rows it sends.  GPU: the step stream waits on the collective's completion event; CPU: the gloo work item.
            pending, grp.inflight[grp.slot] = grp.inflight[grp.slot], None
            if pending is not None:
                if self._cuda:
                    torch.cuda.current_stream(self.device).wait_event(pending)
                else:
                    pending.wait()
        if self.fused:  # this step's rows go into slot (pair grp.slot, step grp.t) — launches take their parameters by value
            grp.env.batch.set_packed_output(grp.packed[grp.slot][grp.t][:grp.n])

    def _after_step(self, grp):
        if not self.fused:
            grp.env.batch.packed_obs(grp.packed[grp.slot][grp.t][:grp.n])
        grp.t += 1
        if grp.t == self.every:
            grp.last, grp.t, grp.slot = grp.slot, 0, grp.slot ^ 1

    def _step(self, name, group, *a, **kw):
        grp = self.groups[group]
        self._before_step(grp)
        out = getattr(grp.env, name)(*a, **kw)
        self._after_step(grp)
        return out

    def step_bbox(self, bbox, op, group=0, next_operation=None):
        """(next_operation: round 4's hint, accepted and ignored — launches order themselves, see ARCVecEnv.step_bbox)"""
        return self._step("step_bbox", group, bbox, op)

    def step_point(self, xy, op, group=0, next_operation=None):
        return self._step("step_point", group, xy, op)

    def step(self, action, group=0):
        return self._step("step", group, action)

    def local_slice(self, global_tensor, group=0):
        """This rank's rows (of ping-pong group `group`) of a [global_envs, ...] tensor, e.g. actions produced by a central policy."""
        grp = self.groups[group]
        return global_tensor[grp.lo:grp.hi]

    def group_global_ids(self, group=0):
        """Global env id of every row `gather(group=...)` returns (int64 [G_group]); for groups == 1 simply arange(global_envs)."""
        return self._ids[group][0]

    # ---- the one collective ----------------------------------------------------------------------------------
    def ready(self, group=0):
        """every = K: True when the K-th step of the current window has run (a gather is due)."""
        return self.groups[group].last is not None

    def gather_async(self, group=0):
        """Starts the all-gather of the rows the last step (the last K steps) of `group` produced, off the step stream: on the GPU
        it is enqueued on a side stream behind an event recorded now, so kernels launched afterwards on the step stream overlap it;
        on the CPU (gloo) it is an async work item.  Returns a GatherWork."""
        grp = self.groups[group]
        if grp.last is None:
            raise RuntimeError(f"gather: {self.every - grp.t} more step(s) of group {group} needed to complete the window of {self.every}")
        slot, grp.last = grp.last, None
        if not self._collective:
            return GatherWork(self, group, slot)
        if self._cuda:
            cur = torch.cuda.current_stream(self.device)
            ev = torch.cuda.Event()
            ev.record(cur)  # the rows are complete here
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                if grp.consumed[slot] is not None:  # whoever read full[slot] two gathers ago has finished
                    self._side.wait_event(grp.consumed[slot])
                dist.all_gather_into_tensor(grp.full[slot], grp.packed[slot], group=self.group)
                done = torch.cuda.Event()
                done.record(self._side)
            grp.inflight[slot] = done
            return GatherWork(self, group, slot, event=done)
        work = dist.all_gather_into_tensor(grp.full[slot], grp.packed[slot], group=self.group, async_op=True)
        grp.inflight[slot] = work
        return GatherWork(self, group, slot, work=work)

    def release(self, work):
        """Tells the double buffer that the caller's reads of `work`'s tensors (enqueued on the current stream so far) are the last
        ones: the gather after the next one may overwrite them behind this point.  Optional on the CPU."""
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.groups[work.group].consumed[work.slot] = ev

    def gather(self, group=0):
        """All ranks receive (grid [G,H,W] int8, grid_dim [G,2] int8, reward [G] int32, terminated [G] bool) of the step that just
        ran — with every = K a leading K axis: [K,G,...] — G = global envs (of the group), rows in global env order: ONE
        all_gather_into_tensor.  Synchronous form of gather_async().wait()."""
        return self.gather_async(group).wait()

    def _unpack(self, group, slot):
        grp = self.groups[group]
        b = grp.env.batch
        K, nm = self.every, self._n_max[group]
        if self._collective:
            rows = grp.full[slot].view(self.world, K, nm, -1).permute(1, 0, 2, 3).reshape(K, self.world * nm, -1)  # rank-major per step
        else:
            rows = grp.packed[slot]
        keep = self._ids[group][1]
        if keep is not None:
            rows = rows[:, keep]
        elif not self._collective and grp.n != nm:
            rows = rows[:, :grp.n]
        G = rows.shape[1]
        grid, gdim, rew, term = EnvBatch.unpack_obs(rows.reshape(K * G, -1), b.H, b.W)
        if K == 1:
            return grid, gdim, rew, term
        return grid.reshape(K, G, b.H, b.W), gdim.reshape(K, G, 2), rew.reshape(K, G), term.reshape(K, G)

    # ---- K steps + K collectives in one hipGraph ----------------------------------------------------------------
    def capture(self, payload, operation=None, form="bbox", group=0):
        """Records K = payload.shape[0] steps of `group`, each followed by its all-gather, into ONE hipGraph (the local env's
        launches and the RCCL collective are both capturable): replay() = one host call for K (step, gather) pairs.  Needs
        every == 1.  Returns an object with replay() -> (grid [K,G,H,W], grid_dim [K,G,2], reward [K,G], terminated [K,G]); the actions
        are read from `payload` / `operation` at replay time."""
        assert self.every == 1 and self._cuda and self.fused, "capture: GPU, fused packed rows, every == 1"
        grp = self.groups[group]
        env, b = grp.env, grp.env.batch
        env._many_ok()
        env._check_many(form, payload, operation)
        K, nm, R = int(payload.shape[0]), self._n_max[group], b.packed_obs_size()
        packed = torch.zeros((K, nm, R), dtype=torch.uint8, device=self.device)
        full = torch.zeros((K, self.world * nm, R), dtype=torch.uint8, device=self.device) if self._collective else None
        reward, term, trunc, dense = env._many_buffers(K)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(K):
                b.set_packed_output(packed[i][:grp.n])
                env._enqueue_steps(form, payload[i:i + 1], None if operation is None else operation[i:i + 1], reward[i:i + 1],
                                   term[i:i + 1], None if trunc is None else trunc[i:i + 1], None if dense is None else dense[i:i + 1])
                if self._collective:
                    dist.all_gather_into_tensor(full[i], packed[i], group=self.group)
        self._before_step(grp)  # (back to the eager path's buffer)
        owner = self

        class _Captured:
            steps = K

            def replay(self_inner):
                g.replay()
                rows = full if full is not None else packed
                keep = owner._ids[group][1]
                if keep is not None:
                    rows = rows[:, keep]
                elif full is None and grp.n != nm:
                    rows = rows[:, :grp.n]
                G = rows.shape[1]
                grid, gdim, rew, tm = EnvBatch.unpack_obs(rows.reshape(K * G, -1), b.H, b.W)
                return grid.reshape(K, G, b.H, b.W), gdim.reshape(K, G, 2), rew.reshape(K, G), tm.reshape(K, G)
        c = _Captured()
        c.graph, c.payload, c.operation = g, payload, operation
        c._keep = (packed, full, reward, term, trunc, dense)  # everything the captured launches write stays alive with the graph
        return c
