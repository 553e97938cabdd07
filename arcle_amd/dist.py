"""Multi-GPU: the env batch is partitioned across ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  Envs are independent — `step` has no cross-env dependency
(/root/reference/arcle/envs/o2arcenv.py:130-151 touches only `self`) — so the data path needs NO collective:
rank g owns the contiguous global env ids [g*n, (g+1)*n).  The only exchange that ever happens is the optional
gather of what a central learner consumes, `(obs, reward, done)`; it is a single all_gather_into_tensor per
field (one-shot, every xGMI link carries one shard) and lives here, outside the step path.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(global_envs, world_size, rank):
    """Contiguous global env ids owned by `rank`; the remainder goes to the lowest ranks."""
    base, rem = divmod(int(global_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seed(seed, global_env_id):
    """Per-env RNG substream keyed by the GLOBAL env id, so results do not depend on the number of GPUs
    (splitmix64 finaliser)."""
    z = (int(seed) ^ (int(global_env_id) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


class ShardedVecEnv:
    """Wraps this rank's local vector env (anything with the ARCVecEnv step/reset interface and `.N`) and adds
    global bookkeeping + the (obs, reward, done) gather.  `local_env_factory(n_local, lo, hi)` builds the local
    env for global ids [lo, hi)."""

    def __init__(self, global_envs, local_env_factory, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.global_envs = int(global_envs)
        self.lo, self.hi = shard_range(global_envs, self.world, self.rank)
        if self.global_envs % self.world != 0:
            raise ValueError("global_envs must be divisible by the number of ranks (all_gather_into_tensor needs equal shards)")
        self.local = local_env_factory(self.hi - self.lo, self.lo, self.hi)
        self.N = self.hi - self.lo

    # local stepping: no communication
    def reset(self, **kw):
        return self.local.reset(**kw)

    def step_bbox(self, bbox, op):
        return self.local.step_bbox(bbox, op)

    def step_point(self, xy, op):
        return self.local.step_point(xy, op)

    def step(self, action):
        return self.local.step(action)

    def local_slice(self, global_tensor):
        """This rank's rows of a [global_envs, ...] tensor (e.g. actions produced by a central policy)."""
        return global_tensor[self.lo:self.hi]

    def gather(self, obs, reward, terminated, keys=("grid", "grid_dim")):
        """All ranks receive the [global_envs, ...] versions of the selected obs fields, reward and done.
        One all_gather_into_tensor per field: shard i lands at rows [i*n, (i+1)*n) — i.e. global env order."""
        out = {}
        for k in keys:
            out[k] = self._all_gather(obs[k])
        return out, self._all_gather(reward), self._all_gather(terminated.to(torch.uint8)).bool()

    def _all_gather(self, t):
        t = t.contiguous()
        if self.world == 1:
            return t
        full = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(full, t, group=self.group)
        return full
