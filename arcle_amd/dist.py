"""Multi-GPU: the env batch is partitioned across ranks (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm).  Envs are independent — `step` has no cross-env dependency
(/root/reference/arcle/envs/o2arcenv.py:130-151 touches only `self`) — so the data path needs NO collective:
rank g owns the contiguous global env ids [g*n, (g+1)*n) and keys its device-side task draws by the GLOBAL env id
(arcle_amd/sampling.py), so the trajectories do not depend on the number of GPUs.  The only exchange that ever happens
is the optional gather of what a central learner consumes, (grid, grid_dim, reward, done): the step kernel itself packs its
outputs into one 912-byte record per env (STEP_PACK_OBS, a fused epilogue; arcle_pack_obs is the stand-alone form) and the
records move with ONE all_gather_into_tensor per step (one-shot, every xGMI link carries one shard).
"""
import torch
import torch.distributed as dist

from .engine import EnvBatch


def shard_range(global_envs, world_size, rank):
    """Contiguous global env ids owned by `rank`; the remainder goes to the lowest ranks."""
    base, rem = divmod(int(global_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedVecEnv:
    """This rank's shard of a global batch of `global_envs` envs.  `local_env_factory(n_local, lo, hi)` builds the local
    vector env for the global ids [lo, hi) — for the HIP path: `lambda n, lo, hi: ARCVecEnv(cls, n, loader, seed=S,
    env_base=lo, ...)` (env_base makes the device-side task draws follow the global env id)."""

    def __init__(self, global_envs, local_env_factory, group=None, fused_pack=True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.global_envs = int(global_envs)
        self.lo, self.hi = shard_range(global_envs, self.world, self.rank)
        if self.global_envs % self.world != 0:
            raise ValueError("global_envs must be divisible by the number of ranks (all_gather_into_tensor needs equal shards)")
        self.local = local_env_factory(self.hi - self.lo, self.lo, self.hi)
        self.N = self.hi - self.lo
        self._packed = self._full = None
        # fused_pack: the local env's step kernel writes the packed rows itself (ARCVecEnv.enable_packed_rows), so that
        # gather() is the collective alone; otherwise gather() launches arcle_pack_obs first
        self.fused = bool(fused_pack) and hasattr(self.local, "enable_packed_rows")
        if self.fused:
            self._packed = self.local.enable_packed_rows()

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

    def gather(self):
        """All ranks receive (grid [G,H,W] int8, grid_dim [G,2] int8, reward [G] int32, terminated [G] bool) of the step
        that just ran, G = global_envs, rows in global env order: ONE all_gather_into_tensor (plus one packing launch when the rows are not written by the step kernel)."""
        b = self.local.batch
        if self._packed is None:
            self._packed = torch.empty((self.N, b.packed_obs_size()), dtype=torch.uint8, device=b.device)
        if self._full is None:
            self._full = self._packed if self.world == 1 else torch.empty(
                (self.world * self.N, self._packed.shape[1]), dtype=torch.uint8, device=b.device)
        if not self.fused:
            b.packed_obs(self._packed)
        if self.world > 1:
            dist.all_gather_into_tensor(self._full, self._packed, group=self.group)  # shard i -> rows [i*n, (i+1)*n)
        return EnvBatch.unpack_obs(self._full, b.H, b.W)
