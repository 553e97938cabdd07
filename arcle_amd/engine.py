"""Device-resident state of a batch of ARCLE envs + the calls into libarcle_hip.so.

`EnvBatch` owns the HBM buffers (torch tensors: plumbing for device memory and streams only) in the
layout include/arcle_hip.h declares and hands their raw pointers to the C ABI.  All stepping happens
in the HIP kernels; this file contains no grid arithmetic.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import ArcleHipError

PLANES = ["input", "grid", "selected", "clip", "object", "object_sel", "background", "answer"]
PLANE_ID = {k: i for i, k in enumerate(PLANES)}
REC_FIELDS = {  # name -> (byte offset, length) inside the 16-byte per-env record
    "input_dim": (0, 2), "grid_dim": (2, 2), "clip_dim": (4, 2), "object_dim": (6, 2), "object_pos": (8, 2),
    "trials_remain": (10, 1), "terminated": (11, 1), "active": (12, 1), "rotation_parity": (13, 1),
    "answer_dim": (14, 2),
}
KIND_PLANES = {  # which planes each env kind's state dict holds (o2arcenv.py:16-34, arcenv.py:81-89)
    "o2arc": PLANES,
    "arc": ["input", "grid", "clip", "answer"],
    "raw": ["input", "grid", "answer"],
}
STEP_AUTORESET = 1
STEP_ELIDE_SELECTED = 2
STEP_TRUNCATE = 4
STEP_RESAMPLE = 8
STEP_DENSE = 16
STEP_CONTINUE_RULE = 32
STEP_RESET_ON_SUBMIT = 64
STEP_FLAT_OBS = 128
STEP_PACK_OBS = 256
STEP_ROWS_INCREMENTAL = 512
AUG_PERMUTE, AUG_ROT90 = 1, 2
ST_BAD_OP, ST_ROTATE_DOMAIN, ST_BAD_TASK, ST_BAD_SELECTION, ST_AUG_DOMAIN = 1, 2, 4, 8, 16
ROW_TAIL = 16  # bytes of the optional step-output tail of a flat row (arcle_set_flat_output_ex)
MAX_CELLS = 1024  # ARCLE_MAX_CELLS: one 64-lane wavefront x 16 cells per lane holds a whole H x W plane (the one-wavefront-per-env kernels)
MAX_SIDE = 127    # dims travel as int8 in the per-env record (and in the reference's state dict, base.py:162-166)


def is_big_grid(H, W):
    """More than ARCLE_MAX_CELLS cells: the handle is served by the workgroup-per-env kernels (arcle_amd/csrc/arcle_big.hip)."""
    return int(H) * int(W) > MAX_CELLS


def check_grid_size(H, W):
    """The reference's constructor takes any max_grid_size (base.py:37-49) and stores the dims as int8 (base.py:162-166), so sides up to
    127 are meaningful.  H * W <= 1024 runs on the one-wavefront-per-env kernels (ARC grids are at most 30 x 30: the headline path);
    larger planes on the workgroup-per-env kernels — the same API minus what `EnvBatch.BIG_UNSUPPORTED` lists.  A side beyond 127 is
    refused here, before any device is touched; arcle_create reports the same as ARCLE_ERR_CONFIG."""
    H, W = int(H), int(W)
    if H <= 0 or W <= 0:
        raise ValueError(f"max_grid_size must be positive, got ({H}, {W})")
    if H > MAX_SIDE or W > MAX_SIDE:
        raise ValueError(f"max_grid_size ({H}, {W}) is outside what the HIP path supports: H, W <= {MAX_SIDE} (grid dims are int8 in the "
                         f"state dict, as in the reference); ARC grids are at most 30 x 30")


_hip = None


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class EnvBatch:
    """n_envs envs of one kind on one GPU."""
    # what a batch of more than 1024 cells per plane (`self.big`) does not offer — the library refuses these calls with ARCLE_ERR_CONFIG
    BIG_UNSUPPORTED = ("autotune (one launch plan: returns no candidates)",)

    def __init__(self, n_envs, H, W, max_trial=-1, kind="o2arc", device=None, plane_stride=None):
        check_grid_size(H, W)
        if not torch.cuda.is_available():
            raise ArcleHipError("no HIP device visible (torch.cuda.is_available() is False); "
                                "arcle_amd has no CPU fallback")
        self.L = _lib.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:  # "cuda" means torch's CURRENT device, not ordinal 0 (multi-GPU ranks)
            self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        self.N, self.H, self.W, self.P = int(n_envs), int(H), int(W), int(H) * int(W)
        self.big = is_big_grid(H, W)
        if plane_stride is None:
            plane_stride = int(os.environ.get("ARCLE_PLANE_STRIDE", "0")) or ((self.P + 127) & ~127)  # ARCLE_DEFAULT_PLANE_STRIDE
        self.PS = int(plane_stride)  # plane stride: one aligned dwordx4 per lane
        self.max_trial, self.kind = int(max_trial), kind
        # 16 B alignment of every plane/record row comes from torch's >=256 B allocation alignment
        self.planes = {k: torch.zeros((self.N, self.PS), dtype=torch.int8, device=self.device)
                       for k in KIND_PLANES[kind]}
        self.rec = torch.zeros((self.N, 16), dtype=torch.int8, device=self.device)
        self.cnt = torch.zeros((self.N, 2), dtype=torch.int32, device=self.device)
        self.reward = torch.zeros(self.N, dtype=torch.int32, device=self.device)
        self.term = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
        cfg = _lib.Config(self.N, self.H, self.W, self.max_trial, self.device.index, self.PS)
        bufs = _lib.Buffers()
        for k, i in PLANE_ID.items():
            bufs.plane[i] = self.planes[k].data_ptr() if k in self.planes else None
        bufs.rec = self.rec.data_ptr()
        bufs.cnt = self.cnt.data_ptr()
        h = ctypes.c_void_p()
        rc = self.L.arcle_create(ctypes.byref(cfg), ctypes.byref(bufs), ctypes.byref(h))
        if rc != 0:
            raise ArcleHipError(f"arcle_create failed with status {rc}")
        self._h = h
        self.bits_stride = int(self.L.arcle_mask_bits_stride(h))  # bytes between the envs' rows of a bit-packed mask array (128 up to 1024 cells)
        self.n_ops = 0
        self._reward_ptr, self._term_ptr = self.reward.data_ptr(), self.term.data_ptr()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.L.arcle_destroy(h)
            self._h = None

    # ---- helpers ------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            msg = self.L.arcle_last_error(self._h)
            raise ArcleHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def _stream(self):
        """Raw handle of torch's CURRENT stream on this device (the private accessor is several times cheaper than building a
        Stream object per call; it is what torch's own inductor runtime uses)."""
        try:
            return torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:  # pragma: no cover - older / newer torch without the private accessor
            return torch.cuda.current_stream(self.device).cuda_stream

    def sync(self, stream=None):
        """Blocks until `stream` (default: torch's current stream on this device) has drained — hipStreamSynchronize called directly
        (the latency path of the single-env class; saves the Python-side bookkeeping of torch's Stream objects)."""
        global _hip
        if _hip is None:
            _hip = ctypes.CDLL("libamdhip64.so")
            _hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
        rc = _hip.hipStreamSynchronize(self._stream() if stream is None else stream)
        if rc != 0:
            raise ArcleHipError(f"hipStreamSynchronize failed ({rc})")

    def next_seq(self, tail_u8):
        """Arms the completion signal of the next launch that writes the row tail `tail_u8` (arcle_set_flat_seq) and returns its
        sequence number.  The tail's signal byte is cleared first: one counter serves several tail buffers (the step() row and the
        transition() row of a single env), so a stale byte of an earlier launch on THIS buffer could equal the new number — the
        launch is stream-ordered after this host write to pinned memory."""
        tail_u8[15] = 0
        self._seq = getattr(self, "_seq", 0) % 255 + 1
        rc = self.L.arcle_set_flat_seq(self._h, self._seq)
        if rc != 0:
            self._check(rc, "arcle_set_flat_seq")
        return self._seq

    def wait_tail(self, tail_u8, seq, stream=None, spins=40000):
        """Waits for the launch armed with `seq` by polling byte 15 of its row tail in PINNED host memory (`tail_u8`: a 16-element uint8
        numpy view): the kernel stores that byte last, behind a system-scope release, so the row is complete when it shows up.  A bounded
        spin (~2 ms); then the stream is synchronised instead (a hung or failed launch surfaces there)."""
        for _ in range(spins):
            if tail_u8[15] == seq:
                return
        self.sync(stream)
        if tail_u8[15] != seq:
            raise ArcleHipError("the step finished without writing its row tail")

    def plane(self, name):
        """Zero-copy [N,H,W] int8 view of a state plane (row stride PS)."""
        return self.planes[name][:, :self.P].view(self.N, self.H, self.W)

    def field(self, name):
        """Zero-copy [N,len] int8 view of a scalar field of the per-env record."""
        off, n = REC_FIELDS[name]
        return self.rec[:, off:off + n]

    # ---- configuration ------------------------------------------------------------------------
    def set_op_table(self, descs):
        arr = (ctypes.c_uint32 * len(descs))(*[int(d) for d in descs])
        self._check(self.L.arcle_set_op_table(self._h, arr, len(descs)), "arcle_set_op_table")
        self.n_ops = len(descs)
        # flag for step launches on states that only ever evolved through this library (see ARCLE_STEP_ELIDE_SELECTED)
        self.elide_flag = STEP_ELIDE_SELECTED if self.L.arcle_can_elide_selected(self._h) else 0

    def set_tasks(self, inputs, answers, env_ids=None):
        """Uploads tasks.  inputs/answers: sequences of un-padded 2-D int8 arrays (what Loader.pick yields)."""
        ids = range(self.N) if env_ids is None else list(env_ids)
        n = len(ids)
        pin = np.zeros((n, self.PS), np.int8)
        pan = np.zeros((n, self.PS), np.int8)
        dims = np.zeros((n, 4), np.int8)
        for j, (a, b) in enumerate(zip(inputs, answers)):
            a = np.asarray(a, np.int8)
            b = np.asarray(b, np.int8)
            if a.shape[0] > self.H or a.shape[1] > self.W or b.shape[0] > self.H or b.shape[1] > self.W:
                raise ValueError("task grid larger than max_grid_size")
            pin[j, :self.P].reshape(self.H, self.W)[:a.shape[0], :a.shape[1]] = a
            pan[j, :self.P].reshape(self.H, self.W)[:b.shape[0], :b.shape[1]] = b
            dims[j] = (a.shape[0], a.shape[1], b.shape[0], b.shape[1])
        idx = torch.as_tensor(list(ids), device=self.device, dtype=torch.long)
        self.planes["input"][idx] = torch.from_numpy(pin).to(self.device)
        self.planes["answer"][idx] = torch.from_numpy(pan).to(self.device)
        d = torch.from_numpy(dims).to(self.device)
        self.rec[idx, 0:2] = d[:, 0:2]
        self.rec[idx, 14:16] = d[:, 2:4]

    def set_tasks_padded(self, inp, input_dim, ans, answer_dim):
        """Uploads already padded [N,H,W] int8 arrays + [N,2] dims (numpy or torch)."""
        t = lambda x: (x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))).to(self.device, torch.int8)  # noqa: E731
        self.plane("input").copy_(t(inp))
        self.plane("answer").copy_(t(ans))
        self.field("input_dim").copy_(t(input_dim))
        self.field("answer_dim").copy_(t(answer_dim))

    def set_task_table(self, inputs, answers):
        """Packs a list of (input, answer) grid pairs (un-padded 2-D int8 arrays) into the device task table that
        `reset_from_table` indexes.  One upload; resets afterwards move no grids over PCIe."""
        T = len(inputs)
        tin = np.zeros((T, self.PS), np.int8)
        tan = np.zeros((T, self.PS), np.int8)
        din = np.zeros((T, 2), np.int8)
        dan = np.zeros((T, 2), np.int8)
        for j, (a, b) in enumerate(zip(inputs, answers)):
            a, b = np.asarray(a, np.int8), np.asarray(b, np.int8)
            if max(a.shape[0], b.shape[0]) > self.H or max(a.shape[1], b.shape[1]) > self.W:
                raise ValueError("task grid larger than max_grid_size")
            tin[j, :self.P].reshape(self.H, self.W)[:a.shape[0], :a.shape[1]] = a
            tan[j, :self.P].reshape(self.H, self.W)[:b.shape[0], :b.shape[1]] = b
            din[j], dan[j] = a.shape, b.shape
        self._table = [torch.from_numpy(x).to(self.device) for x in (tin, din, tan, dan)]  # keep alive
        self._check(self.L.arcle_set_task_table(self._h, _ptr(self._table[0]), _ptr(self._table[1]), _ptr(self._table[2]),
                                                _ptr(self._table[3]), T), "arcle_set_task_table")
        self.n_tasks = T

    def reset_from_table(self, task_idx, mask=None, aug_k=None, aug_perm=None):
        """task_idx: int32 [N] (device) indices into the task table; mask: optional uint8/bool [N].
        aug_k uint8 [N] (np.rot90 count) / aug_perm uint8 [N,10] (colour permutation): explicit augmentation per env
        (agents/env.py:31-42 of the reference)."""
        if task_idx.dtype != torch.int32 or task_idx.device != self.device or not task_idx.is_contiguous():
            task_idx = task_idx.to(device=self.device, dtype=torch.int32).contiguous()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        if aug_k is None and aug_perm is None:
            self._check(self.L.arcle_reset_from_table(self._h, _ptr(task_idx), _ptr(m), self._stream()), "arcle_reset_from_table")
            return
        k = None if aug_k is None else torch.as_tensor(aug_k, device=self.device).to(torch.uint8).contiguous()
        pm = None
        if aug_perm is not None:
            pm = torch.zeros((self.N, 16), dtype=torch.uint8, device=self.device)
            pm[:, :10] = torch.as_tensor(aug_perm, device=self.device).to(torch.uint8)
        self._check(self.L.arcle_reset_from_table_aug(self._h, _ptr(task_idx), _ptr(m), _ptr(k), _ptr(pm), self._stream()),
                    "arcle_reset_from_table_aug")
        torch.cuda.current_stream(self.device).synchronize()  # (k / pm are temporaries)

    def set_sampler(self, pair_off, pair_cnt, seed, env_base=0, aug_flags=0):
        """Device-side task choice (arcle_set_sampler): pair_off / pair_cnt = first table entry / number of entries of
        every candidate problem.  Draws are keyed by (seed, env_base + env, episode)."""
        self._pair = [torch.as_tensor(np.asarray(x, np.int32)).to(self.device) for x in (pair_off, pair_cnt)]  # keep alive
        assert int(self._pair[1].min()) > 0, "every candidate problem needs at least one pair"
        if not hasattr(self, "episode"):
            self.episode = torch.zeros(self.N, dtype=torch.int32, device=self.device)
            self.cur_task = torch.full((self.N,), -1, dtype=torch.int32, device=self.device)
        self._check(self.L.arcle_set_sampler(self._h, _ptr(self._pair[0]), _ptr(self._pair[1]), len(pair_cnt),
                                             ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), ctypes.c_int64(int(env_base)),
                                             _ptr(self.episode), _ptr(self.cur_task), int(aug_flags)), "arcle_set_sampler")

    def reset_sampled(self, mask=None):
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self._check(self.L.arcle_reset_sampled(self._h, _ptr(m), self._stream()), "arcle_reset_sampled")

    def set_truncation(self, step_limit):
        """ARCLE_STEP_TRUNCATE output: self.trunc[env] = action_steps >= step_limit (gymnasium TimeLimit)."""
        self.trunc = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
        self._check(self.L.arcle_set_truncation(self._h, _ptr(self.trunc), int(step_limit)), "arcle_set_truncation")

    def set_dense_output(self):
        """ARCLE_STEP_DENSE output: self.dense int32 [N,2] = (matching cells, total cells) of agents/env.py:44-58."""
        self.dense = torch.zeros((self.N, 2), dtype=torch.int32, device=self.device)
        self._check(self.L.arcle_set_dense_output(self._h, _ptr(self.dense)), "arcle_set_dense_output")

    # ---- the hot path -------------------------------------------------------------------------
    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self._check(self.L.arcle_reset(self._h, _ptr(m), self._stream()), "arcle_reset")

    def _step(self, fn, payload, op, flags):
        if op.dtype != torch.int32 or not op.is_contiguous() or op.device != self.device:
            op = op.to(device=self.device, dtype=torch.int32).contiguous()
        # (plain ints for the pointer arguments: ctypes converts them without a c_void_p object per argument)
        rc = fn(self._h, payload.data_ptr(), op.data_ptr(), self._reward_ptr, self._term_ptr, flags, self._stream())
        if rc != 0:
            self._check(rc, "arcle_step")
        return self.reward, self.term

    def step_bbox(self, bbox, op, flags=0):
        """bbox int32 [N,4] = (x1,y1,x2,y2) as BBoxWrapper.action (bbox.py:22-30); op int32 [N]."""
        if bbox.dtype != torch.int32 or not bbox.is_contiguous() or bbox.device != self.device:
            bbox = bbox.to(device=self.device, dtype=torch.int32).contiguous()
        return self._step(self.L.arcle_step_bbox, bbox, op, flags)

    def step_point(self, xy, op, flags=0):
        if xy.dtype != torch.int32 or not xy.is_contiguous() or xy.device != self.device:
            xy = xy.to(device=self.device, dtype=torch.int32).contiguous()
        return self._step(self.L.arcle_step_point, xy, op, flags)

    def step_mask(self, sel, op, flags=0):
        """sel [N,H,W] int8/bool selection masks (action['selection'])."""
        if sel.dtype == torch.bool:
            sel = sel.to(torch.int8)
        if sel.dtype != torch.int8 or not sel.is_contiguous() or sel.device != self.device:
            sel = sel.to(device=self.device, dtype=torch.int8).contiguous()
        return self._step(self.L.arcle_step_mask, sel, op, flags)

    def step_bbox5(self, act5, flags=0):
        """act5 int32 [N,5] = the BBoxWrapper action (x1, y1, x2, y2, operation) as one record per env (bbox.py:22-30): no separate
        op array.  `act5` may also be PINNED HOST memory (the kernel reads it over PCIe: no copy node in front of the step)."""
        if act5.dtype != torch.int32 or not act5.is_contiguous() or not (act5.device == self.device or act5.is_pinned()):
            act5 = act5.to(device=self.device, dtype=torch.int32).contiguous()
        self._check(self.L.arcle_step_bbox5(self._h, _ptr(act5), _ptr(self.reward), _ptr(self.term), int(flags), self._stream()),
                    "arcle_step_bbox5")
        return self.reward, self.term

    def step_bits(self, bits, op, flags=0):
        """bits uint8 [N,128]: bit-packed boolean selection masks (bit f of row e = cell f of env e; `pack_mask_bits` makes them)."""
        assert bits.dtype == torch.uint8 and bits.shape == (self.N, self.bits_stride) and bits.is_contiguous() and bits.device == self.device
        return self._step(self.L.arcle_step_bits, bits, op, flags)

    def pack_mask_bits(self, sel, out=None):
        """int8/bool [N,H,W] selection masks -> uint8 [N,128] bit-packed rows (one launch)."""
        if sel.dtype == torch.bool:
            sel = sel.view(torch.int8) if sel.is_contiguous() else sel.to(torch.int8)
        if sel.dtype != torch.int8 or not sel.is_contiguous() or sel.device != self.device:
            sel = sel.to(device=self.device, dtype=torch.int8).contiguous()
        if out is None:
            out = torch.empty((self.N, self.bits_stride), dtype=torch.uint8, device=self.device)
        self._check(self.L.arcle_pack_mask_bits(self._h, _ptr(sel), _ptr(out), self._stream()), "arcle_pack_mask_bits")
        return out

    def step_many(self, form, payload, op=None, flags=0, reward=None, term=None):
        """K step() launches enqueued by ONE call into the library (arcle_step_many): payload [K, N, ...] in the ingress form
        `form` ("mask" | "bbox" | "point" | "bbox5" | "bits"), op int32 [K, N] (None for "bbox5") -> (reward int32 [K, N],
        terminated uint8 [K, N]).  Every step is a full step; self.reward / self.term are not written."""
        K = int(payload.shape[0])
        # ("bbox5" records may live in PINNED HOST memory: the library then prefetches step t+1's records under launch t)
        assert payload.is_contiguous() and (payload.device == self.device or (form == "bbox5" and payload.is_pinned()))
        assert op is None or (op.is_contiguous() and op.dtype == torch.int32)
        if reward is None:
            reward = torch.empty((K, self.N), dtype=torch.int32, device=self.device)
        if term is None:
            term = torch.empty((K, self.N), dtype=torch.uint8, device=self.device)
        self._check(self.L.arcle_step_many(self._h, _lib.INGRESS[form], K, _ptr(payload), _ptr(op), _ptr(reward), _ptr(term), int(flags),
                                           self._stream()), "arcle_step_many")
        return reward, term

    def set_dispatch_order(self, enable=True):
        """Dispatch order on / off for this handle (arcle_set_dispatch_order).  On (default): launches of the standard 30 x 30 batch
        hand the object operations (Move / Rotate / Flip, the longest waves) to the waves that start first — every launch derives that
        from the operations it is about to execute, inside groups of 32 envs; scheduling only, results never depend on it."""
        self._check(self.L.arcle_set_dispatch_order(self._h, 1 if enable else 0), "arcle_set_dispatch_order")

    def launch_info(self, form, flags):
        """The plan a step launch with this ingress form and these flags takes (arcle_launch_info): dict(orders_itself, policy ('' or the
        letter A / B / H / J), waves_per_workgroup, autotuned)."""
        import numpy as np
        out = np.zeros(4, np.int32)
        self._check(self.L.arcle_launch_info(self._h, _lib.INGRESS[form], int(flags), out.ctypes.data), "arcle_launch_info")
        return {"orders_itself": bool(out[0]), "policy": chr(out[1]) if out[1] else "", "waves_per_workgroup": int(out[2]), "autotuned": bool(out[3])}

    def orders_itself(self, form, flags):
        return self.launch_info(form, flags)["orders_itself"]

    def autotune(self, form, payload, op, flags):
        """Times every launch plan this handle can take on the caller's own action tensors — payload [K, N, ...] / op [K, N]: K consecutive
        action batches, a representative stretch of the policy's output (a single [N, ...] batch is accepted, but one repeated batch is
        a poor sample) — and keeps the fastest for later (form, flags) launches (arcle_autotune: state saved and restored, stream
        synchronised).  Returns the candidates as a list of dicts sorted by time."""
        import numpy as np
        rep = np.zeros((16, 4), np.int32)
        K = int(payload.shape[0]) if (op is not None and op.dim() == 2) or (op is None and payload.dim() == 3) else 1
        assert payload.is_contiguous() and (op is None or op.is_contiguous())
        rc = self.L.arcle_autotune(self._h, _lib.INGRESS[form], K, _ptr(payload), _ptr(op), int(flags), rep.ctypes.data, 16, self._stream())
        if rc < 0:
            self._check(rc, "arcle_autotune")
        rows = [{"orders_itself": bool(r[0]), "policy": chr(r[1]) if r[1] else "", "waves_per_workgroup": int(r[2]), "us_per_launch": r[3] / 1e3} for r in rep[:rc]]
        return sorted(rows, key=lambda r: r["us_per_launch"])

    def hint_next_ops(self, next_op, stride=1):
        """ABI 4's one-shot hint of the NEXT step's operations.  Launches order themselves since ABI 5, so this does nothing; kept so
        that callers written against it keep running."""
        return None

    def step_bbox_ptr(self, bbox_ptr, op_ptr, flags=0, stream=0):
        """Lowest-overhead launch for rollout loops: raw device addresses (ints) of an int32 [N,4] bbox array
        and an int32 [N] op array, explicit stream handle.  Outputs land in self.reward / self.term."""
        rc = self.L.arcle_step_bbox(self._h, bbox_ptr, op_ptr, self._reward_ptr, self._term_ptr, flags, stream)
        if rc != 0:
            self._check(rc, "arcle_step_bbox")

    def rollout(self, payload, op, flags=0, point=False, mask=False, packed=None):
        """T steps in one launch.  payload int32 [T,N,4] (bbox) / [T,N,2] (point) / int8 [T,N,H,W] (mask=True), op
        int32 [T,N]; returns (reward int32 [T,N], terminated uint8 [T,N]).  Same semantics as T step_* calls.
        packed: a uint8 [T, N, packed_obs_size()] device tensor — the launch then also writes the packed observation row of EVERY step
        (grid | grid_dim | reward | terminated; STEP_PACK_OBS; unpack with EnvBatch.unpack_obs): an observation after every step although
        the state never leaves the chip between the steps."""
        T = int(op.shape[0])
        if packed is not None:
            assert packed.shape == (T, self.N, self.packed_obs_size()) and packed.dtype == torch.uint8 and packed.is_contiguous() and packed.device == self.device
            prev = getattr(self, "packed", None)
            self._check(self.L.arcle_set_packed_output(self._h, _ptr(packed)), "arcle_set_packed_output")
            try:
                return self.rollout(payload, op, int(flags) | STEP_PACK_OBS, point, mask)
            finally:
                self._check(self.L.arcle_set_packed_output(self._h, _ptr(prev)), "arcle_set_packed_output")
        op = op.to(device=self.device, dtype=torch.int32).contiguous()
        if mask:
            payload = payload.to(device=self.device, dtype=torch.int8).contiguous()
            assert payload.shape == (T, self.N, self.H, self.W)
        else:
            payload = payload.to(device=self.device, dtype=torch.int32).contiguous()
            assert payload.shape == (T, self.N, 2 if point else 4)
        assert op.shape == (T, self.N)
        reward = torch.empty((T, self.N), dtype=torch.int32, device=self.device)
        term = torch.empty((T, self.N), dtype=torch.uint8, device=self.device)
        fn = self.L.arcle_rollout_mask if mask else (self.L.arcle_rollout_point if point else self.L.arcle_rollout_bbox)
        self._check(fn(self._h, T, _ptr(payload), _ptr(op), _ptr(reward), _ptr(term), int(flags), self._stream()),
                    "arcle_rollout")
        return reward, term

    def flat_obs_size(self, filtered=False):
        """Logical row length of the flattened observation (7*H*W + 14 for O2ARCv2Env; 3*H*W + 10 filtered)."""
        n = self.L.arcle_flat_obs_size(self._h, int(filtered))
        if n < 0:
            raise ArcleHipError("the FilterO2ARC subset needs the O2ARCv2Env state planes")
        return n

    def _flat_buffer(self, filtered):
        L = self.flat_obs_size(filtered)
        return torch.empty((self.N, (L + 15) & ~15), dtype=torch.int8, device=self.device), L

    def flat_obs(self, out=None, filtered=False):
        """[N, L] int8 flattened observations in Gymnasium FlattenObservation key order (agents/models/GPTPolicy.py:
        17-35 of the reference), or the FilterO2ARC subset (agents/env.py:109-126).  Returned as a view of a [N, stride]
        buffer (stride = L rounded up to 16) so that every row store is aligned; pass that buffer back as `out` to reuse it."""
        L = self.flat_obs_size(filtered)
        if out is None:
            out, _ = self._flat_buffer(filtered)
        assert out.dtype == torch.int8 and out.device == self.device and out.shape == (self.N, (L + 15) & ~15) and out.is_contiguous()
        self._check(self.L.arcle_flatten_obs(self._h, _ptr(out), out.shape[1], int(filtered), self._stream()), "arcle_flatten_obs")
        return out[:, :L]

    def set_flat_output(self, filtered=False, tail=False, host=False):
        """Installs the destination of STEP_FLAT_OBS: every step call with that flag also refreshes `self.flat` ([N, L]).
        tail=True: each row's stride ends with 16 bytes of step outputs (`self.flat_tail` int32 [N, 4] view: reward, action_steps,
        submit_count, terminated | truncated << 8 | status << 16).  host=True: the rows live in PINNED HOST memory — the step kernel
        writes them across PCIe itself, no device->host copy afterwards (single envs / small batches)."""
        L = self.flat_obs_size(filtered)
        stride = ((L + 15) & ~15) + (ROW_TAIL if tail else 0)
        if host:
            self._flat_buf = torch.zeros((self.N, stride), dtype=torch.int8).pin_memory()
        else:
            self._flat_buf = torch.zeros((self.N, stride), dtype=torch.int8, device=self.device)
        self._check(self.L.arcle_set_flat_output_ex(self._h, _ptr(self._flat_buf), stride, int(filtered), int(tail)),
                    "arcle_set_flat_output_ex")
        self.flat = self._flat_buf[:, :L]
        self.flat_tail = self._flat_buf[:, stride - ROW_TAIL:].view(torch.int32) if tail else None
        return self.flat

    # ---- state rows: checkpoint / restore and the stateless batched transition ------------------------------------------------
    def state_row_size(self):
        return self.flat_obs_size(False)

    def get_state_rows(self, out=None):
        """The state dict of every env as one row [N, L] int8 (full FlattenObservation layout) — see set_state_rows."""
        return self.flat_obs(out, False)

    def set_state_rows(self, rows, mask=None):
        """rows int8 [N, >= L] (device or pinned host; any stride) -> the resident state of the (masked) envs: the inverse of
        get_state_rows.  The task side (answer, answer_dim) and the counters are not part of a row and stay."""
        assert rows.dtype == torch.int8 and rows.dim() == 2 and rows.shape[0] == self.N and rows.stride(1) == 1
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        self._check(self.L.arcle_set_state_rows(self._h, _ptr(rows), rows.stride(0), _ptr(m), self._stream()), "arcle_set_state_rows")
        if m is not None:
            torch.cuda.current_stream(self.device).synchronize()  # (m may be a temporary)

    def transition_rows(self, rows, form, payload, op, src_env=None, out=None, tail=False, flags=0, reward=None, term=None):
        """`transition(state, action)` of the reference (o2arcenv.py:149-151) for a batch of M (state row, action) pairs, none of
        which touches the resident envs: rows int8 [M, >= L]; payload in the ingress form `form` ("mask" int8 [M,H,W] | "bbox" int32
        [M,4] | "point" int32 [M,2]); op int32 [M]; src_env int32 [M] = the resident env whose answer a Submit / the reward compares
        with (None: env r for row r).  Returns (rows_out [M, stride], reward int32 [M], terminated uint8 [M]); with tail=True the last
        16 bytes of every output row carry (reward, 1, submit counted, terminated | status << 16).  All arrays device or pinned host."""
        M = int(rows.shape[0])
        L = self.flat_obs_size(False)
        stride = ((L + 15) & ~15) + (ROW_TAIL if tail else 0)
        if out is None:
            out = torch.empty((M, stride), dtype=torch.int8, device=self.device)
        # (`out` may be the buffer `rows` views: rows_out == rows_in with equal strides is the in-place form — untouched planes stay)
        assert out.dim() == 2 and out.shape[0] == M and out.shape[1] >= stride and out.shape[1] % 16 == 0 and out.dtype == torch.int8 and out.is_contiguous()
        stride = out.shape[1]
        assert rows.dtype == torch.int8 and rows.stride(1) == 1 and payload.is_contiguous() and op.dtype == torch.int32 and op.is_contiguous()
        if reward is None:
            reward = torch.empty(M, dtype=torch.int32, device=self.device)
        if term is None:
            term = torch.empty(M, dtype=torch.uint8, device=self.device)
        self._check(self.L.arcle_transition_rows(self._h, M, _ptr(rows), rows.stride(0), _lib.INGRESS[form], _ptr(payload), _ptr(op),
                                                 _ptr(src_env), _ptr(out), stride, int(tail), _ptr(reward), _ptr(term), int(flags),
                                                 self._stream()), "arcle_transition_rows")
        return out, reward, term

    def get_plane(self, name, out=None):
        """One key of the state dict as a dense [N, H, W] int8 array (device tensor, or a pinned host tensor passed as `out`):
        arcle_get_plane, a strided copy on the current stream."""
        if out is None:
            out = torch.empty((self.N, self.H, self.W), dtype=torch.int8, device=self.device)
        assert out.dtype == torch.int8 and out.is_contiguous() and out.numel() == self.N * self.P
        self._check(self.L.arcle_get_plane(self._h, PLANE_ID[name], _ptr(out), self._stream()), "arcle_get_plane")
        return out

    def set_plane(self, name, src):
        """The inverse: dense [N, H, W] int8 (device or pinned host) -> the plane (padding untouched); the caches derived from the
        state are dropped (arcle_invalidate)."""
        assert src.dtype == torch.int8 and src.is_contiguous() and src.numel() == self.N * self.P
        self._check(self.L.arcle_set_plane(self._h, PLANE_ID[name], _ptr(src), self._stream()), "arcle_set_plane")
        self.invalidate()

    def get_state(self):
        """Checkpoint of everything that defines the batch's future: a dict of CLONED device tensors (state planes incl. the task's
        input / answer, the per-env record, the counters, and — when a sampler is installed — the per-env episode numbers and
        current task indices).  `set_state` restores it; trajectories continue bit-identically."""
        st = {"planes": {k: v.clone() for k, v in self.planes.items()}, "rec": self.rec.clone(), "cnt": self.cnt.clone()}
        if hasattr(self, "episode"):
            st["episode"], st["cur_task"] = self.episode.clone(), self.cur_task.clone()
        return st

    def invalidate(self):
        """After editing state planes behind the library's back (plain tensor writes into `planes`): drop what the library derived
        from the old state (the dense-pair cache).  Rows kept by ARCLE_STEP_ROWS_INCREMENTAL must be rewritten by the caller."""
        self._check(self.L.arcle_invalidate(self._h, self._stream()), "arcle_invalidate")

    def set_state(self, st):
        self.invalidate()
        for k, v in st["planes"].items():
            self.planes[k].copy_(v)
        self.rec.copy_(st["rec"])
        self.cnt.copy_(st["cnt"])
        if "episode" in st and hasattr(self, "episode"):
            self.episode.copy_(st["episode"])
            self.cur_task.copy_(st["cur_task"])

    def packed_obs_size(self):
        return int(self.L.arcle_packed_obs_size(self._h))

    def packed_obs(self, out=None):
        """[N, R] uint8 rows = grid | grid_dim | reward (int32 LE) | terminated | padding, R = arcle_packed_obs_size()
        (912 for 30x30): the record a central learner gathers per step (one all-gather, arcle_amd.dist)."""
        R = self.L.arcle_packed_obs_size(self._h)
        if out is None:
            out = torch.empty((self.N, R), dtype=torch.uint8, device=self.device)
        assert out.shape == (self.N, R) and out.dtype == torch.uint8 and out.is_contiguous() and out.device == self.device
        self._check(self.L.arcle_pack_obs(self._h, _ptr(self.reward), _ptr(self.term), _ptr(out), self._stream()),
                    "arcle_pack_obs")
        return out

    def set_packed_output(self, out=None):
        """Installs the destination of STEP_PACK_OBS: every step call with that flag also writes `self.packed` ([N, R] uint8 rows
        grid | grid_dim | reward | terminated) from inside the step kernel — no packing launch before the multi-GPU gather.
        `out`: a caller-owned contiguous uint8 [N, R] device tensor (e.g. one slot of a double buffer); launches already enqueued
        keep writing where they were told to (parameters are taken by value per launch)."""
        if out is None:
            out = torch.empty((self.N, self.packed_obs_size()), dtype=torch.uint8, device=self.device)
        assert out.shape == (self.N, self.packed_obs_size()) and out.dtype == torch.uint8 and out.is_contiguous() and out.device == self.device
        self.packed = out
        self._check(self.L.arcle_set_packed_output(self._h, _ptr(self.packed)), "arcle_set_packed_output")
        return self.packed

    def packed_obs_ptr(self, out_ptr, stream=0):
        """Lowest-overhead form for rollout loops: raw device address of the [N, packed_obs_size()] uint8 output."""
        rc = self.L.arcle_pack_obs(self._h, self._reward_ptr, self._term_ptr, out_ptr, stream)
        if rc != 0:
            self._check(rc, "arcle_pack_obs")

    @staticmethod
    def unpack_obs(rows, H, W):
        """(grid int8 [M,H,W], grid_dim int8 [M,2], reward int32 [M], terminated bool [M]) views/copies of packed rows."""
        P = H * W
        grid = rows[:, :P].view(torch.int8).reshape(-1, H, W)
        gdim = rows[:, P:P + 2].view(torch.int8)
        rew = rows[:, P + 2:P + 6].contiguous().view(torch.int32).reshape(-1)
        term = rows[:, P + 6] != 0
        return grid, gdim, rew, term

    # ---- status / accounting ------------------------------------------------------------------
    def status(self, clear=True):
        s = ctypes.c_uint32(0)
        self._check(self.L.arcle_get_status(self._h, ctypes.byref(s), int(clear), self._stream()), "arcle_get_status")
        return s.value

    def enable_accounting(self, on=True):
        self._check(self.L.arcle_enable_accounting(self._h, int(on)), "arcle_enable_accounting")

    def accounting_ex(self, clear=True):
        """(algorithmic bytes, issued bytes, env-steps) since the last clear — arcle_get_accounting_ex."""
        b, i, s = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.L.arcle_get_accounting_ex(self._h, ctypes.byref(b), ctypes.byref(i), ctypes.byref(s), int(clear), self._stream()),
                    "arcle_get_accounting_ex")
        return b.value, i.value, s.value

    def accounting(self, clear=True):
        b, s = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self.L.arcle_get_accounting(self._h, ctypes.byref(b), ctypes.byref(s), int(clear), self._stream()),
                    "arcle_get_accounting")
        return b.value, s.value
