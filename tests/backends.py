"""Test-side adapters giving the oracle, the wave emulator and the HIP library ONE interface, plus the
golden-fixture replayer.  Test infrastructure only.

Backend interface (all arrays NumPy on the host side of the interface):
    N, H, W, fields                       configuration / list of state field names
    set_tasks(input, input_dim, answer, answer_dim)   padded [N,H,W] + [N,2]
    reset(mask=None)
    step(ingress, payload, op, flags=0) -> (reward int32[N], term uint8[N])   ingress in {'bbox','point','mask'}
    get(field) -> np.int8 array [N,...];  counters() -> int32 [N,2];  status(clear=True) -> int
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
PLANES = O.PLANES
REC = O.REC


# ---- checksum used by the golden fixtures (tests/golden/make_golden.py) ----------------------------
def _splitmix_weights(n, seed=0xC0FFEE):
    out, s = [], seed
    M = 0xFFFFFFFFFFFFFFFF
    for _ in range(n):
        s = (s + 0x9E3779B97F4A7C15) & M
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out.append((z ^ (z >> 31)) | 1)
    return np.array(out, np.uint64)


_W = _splitmix_weights(16384)  # (fixtures of at most 1024 cells use the first 1024: the stream is a prefix)


def checksum(a):
    a = np.ascontiguousarray(a)
    flat = a.reshape(a.shape[0], -1).view(np.uint8).astype(np.uint64) + np.uint64(1)
    with np.errstate(over="ignore"):
        return (flat * _W[: flat.shape[1]]).sum(axis=1, dtype=np.uint64)


BITS_STRIDE = 128


def pack_bits(masks):
    """[N,H,W] masks (truthy = non-zero) -> uint8 [N,128] rows, bit f of a row = cell f (the INGRESS_BITS form of include/arcle_hip.h)."""
    m = np.asarray(masks)
    n = m.shape[0]
    flat = (m.reshape(n, -1) != 0)
    P = flat.shape[1]
    out = np.zeros((n, BITS_STRIDE if P <= 1024 else ((P + 127) & ~127) // 8), np.uint8)  # (arcle_mask_bits_stride: plane stride / 8 beyond 1024 cells)
    pk = np.packbits(flat, axis=1, bitorder="little")
    out[:, :pk.shape[1]] = pk
    return out


def row_layout(kind, P):
    """(field, length) of the full flattened state row in FlattenObservation order (arcle_wave.h flat_row; pinned on the reference's
    FlattenObservation rows by features.flat)."""
    lay = []
    if kind != "raw":
        lay += [("clip", P), ("clip_dim", 2)]
    lay += [("grid", P), ("grid_dim", 2), ("input", P), ("input_dim", 2)]
    if kind == "o2arc":
        lay += [("active", 1), ("background", P), ("object", P), ("object_dim", 2), ("object_pos", 2), ("object_sel", P),
                ("rotation_parity", 1), ("selected", P)]
    lay += [("terminated", 1), ("trials_remain", 1)]
    return lay


def state_rows(be):
    """The full flattened state rows of a backend, built on the host from its fields (int8 [N, L])."""
    parts = [np.asarray(be.get(f)).reshape(be.N, -1) for f, _ in row_layout(be.kind, be.H * be.W)]
    return np.ascontiguousarray(np.concatenate(parts, 1).astype(np.int8))


# ---- oracle ------------------------------------------------------------------------------------------
class OracleBackend:
    name = "oracle"

    def __init__(self, N, H, W, max_trial, kind, ops):
        self.N, self.H, self.W, self.kind = N, H, W, kind
        self.env = O.OracleEnv(N, H, W, max_trial, kind, ops)

    def set_tasks(self, inp, idim, ans, adim):
        self.env.planes["input"][:] = inp
        self.env.planes["answer"][:] = ans
        self.env.field("input_dim")[:] = idim
        self.env.field("answer_dim")[:] = adim

    def reset(self, mask=None):
        self.env.reset(mask)

    def step(self, ingress, payload, op, flags=0):
        if ingress == "bbox5":  # (the oracle knows the three classic forms: the record form is bbox + op)
            payload = np.asarray(payload)
            ingress, payload, op = "bbox", payload[:, :4], payload[:, 4]
        fn = {"bbox": self.env.step_bbox, "point": self.env.step_point, "mask": self.env.step_mask}[ingress]
        r, t = fn(payload, op, flags)
        return r.copy(), t.copy()

    def get(self, field):
        return (self.env.planes[field] if field in self.env.planes else self.env.field(field)).copy()

    def counters(self):
        return self.env.cnt.copy()

    def status(self, clear=True):
        return self.env.status(clear)


# ---- wave emulator (tests/emu/wave_emu.cpp: the kernel body of arcle_wave.h run lock-step on the CPU) ----
class _StepParams(ctypes.Structure):  # mirror of arcle::StepParams (arcle_amd/csrc/arcle_wave.h)
    _fields_ = [("plane", ctypes.c_void_p * 8), ("rec", ctypes.c_void_p), ("cnt", ctypes.c_void_p),
                ("op", ctypes.c_void_p), ("sel", ctypes.c_void_p), ("reward", ctypes.c_void_p),
                ("term", ctypes.c_void_p),
                ("n_envs", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("P", ctypes.c_int32),
                ("PS", ctypes.c_int32), ("n_ops", ctypes.c_int32), ("max_trial", ctypes.c_int32),
                ("ingress", ctypes.c_int32), ("flags", ctypes.c_uint32), ("div_magic", ctypes.c_uint32),
                ("nseg", ctypes.c_int32), ("n_steps", ctypes.c_int32), ("flat_seq", ctypes.c_int32),
                ("step_limit", ctypes.c_int32),
                ("status", ctypes.c_void_p), ("acct", ctypes.c_void_p), ("d_ops", ctypes.c_void_p),
                ("trunc", ctypes.c_void_p), ("dense", ctypes.c_void_p), ("flat_out", ctypes.c_void_p),
                ("flat_stride", ctypes.c_int32), ("flat_filter", ctypes.c_int32), ("pack_out", ctypes.c_void_p),
                ("rmask", ctypes.c_void_p), ("task_idx", ctypes.c_void_p), ("tbl_in", ctypes.c_void_p),
                ("tbl_ans", ctypes.c_void_p), ("tbl_in_dim", ctypes.c_void_p), ("tbl_ans_dim", ctypes.c_void_p),
                ("n_tasks", ctypes.c_int32), ("aug_flags", ctypes.c_uint32), ("seed", ctypes.c_uint64),
                ("env_base", ctypes.c_int64), ("episode", ctypes.c_void_p), ("cur_task", ctypes.c_void_p),
                ("pair_off", ctypes.c_void_p), ("pair_cnt", ctypes.c_void_p), ("aug_k", ctypes.c_void_p),
                ("aug_perm", ctypes.c_void_p), ("n_problems", ctypes.c_int32),
                ("wpw", ctypes.c_int32), ("flat_tail", ctypes.c_int32), ("rows_in", ctypes.c_void_p),
                ("rows_in_stride", ctypes.c_int32), ("n_resident", ctypes.c_int32), ("dense_cache", ctypes.c_void_p),
                ("next_sel", ctypes.c_void_p), ("stage_out", ctypes.c_void_p),
                ("long_mask", ctypes.c_uint64), ("group_magic", ctypes.c_uint32), ("spec_grid", ctypes.c_int32)]


_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        d = os.path.join(ROOT, "tests", "emu")
        so, src = os.path.join(d, "libwave_emu.so"), os.path.join(d, "wave_emu.cpp")
        hdr = os.path.join(ROOT, "arcle_amd", "csrc", "arcle_wave.h")
        if os.environ.get("ARCLE_WAVE_EMU_LIB"):  # (tests/test_emu_sanitized.py: the ASan + UBSan build)
            so = os.environ["ARCLE_WAVE_EMU_LIB"]
        elif not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
        _emu = ctypes.CDLL(so)
        _emu.emu_run.argtypes = [ctypes.c_int, ctypes.POINTER(_StepParams)]
        assert _emu.emu_params_size() == ctypes.sizeof(_StepParams), "StepParams layout drifted"
    return _emu


class EmuBackend:
    name = "emu"
    INGRESS = {"mask": 0, "bbox": 1, "point": 2, "bbox5": 3, "bits": 4}

    PLANE_STRIDE = None  # override: bytes between envs of a plane (default = the library's: H*W rounded up to 128)

    def __init__(self, N, H, W, max_trial, kind, ops):
        self.N, self.H, self.W, self.kind = N, H, W, kind
        self.P = H * W
        self.PS = self.PLANE_STRIDE or ((self.P + 127) & ~127)
        self.max_trial = max_trial
        self.buf = {k: np.zeros((N, self.PS), np.int8) for k in O.KIND_PLANES[kind]}
        self.rec = np.zeros((N, 16), np.int8)
        self.cnt = np.zeros((N, 2), np.int32)
        self.reward = np.zeros(N, np.int32)
        self.term = np.zeros(N, np.uint8)
        self.stat = np.zeros(1, np.uint32)
        self.acct = np.zeros(2 * N, np.uint32)  # [0, N) algorithmic bytes, [N, 2N) issued bytes
        self.ops = list(ops)

    def _params(self):
        p = _StepParams()
        for i, k in enumerate(PLANES):
            p.plane[i] = self.buf[k].ctypes.data if k in self.buf else None
        p.rec, p.cnt = self.rec.ctypes.data, self.cnt.ctypes.data
        p.reward, p.term = self.reward.ctypes.data, self.term.ctypes.data
        p.status, p.acct = self.stat.ctypes.data, self.acct.ctypes.data
        p.n_envs, p.H, p.W, p.max_trial, p.n_ops = self.N, self.H, self.W, self.max_trial, len(self.ops)
        p.PS = self.PS
        self._ops_arr = np.zeros(65, np.uint32)  # the "device" op table (ARCLE_MAX_OPS + 1 slots)
        self._ops_arr[:len(self.ops)] = self.ops
        p.d_ops = self._ops_arr.ctypes.data
        return p

    def plane(self, k):
        return self.buf[k][:, :self.P].reshape(self.N, self.H, self.W)

    def set_tasks(self, inp, idim, ans, adim):
        self.buf["input"][:, :self.P] = np.asarray(inp).reshape(self.N, self.P)
        self.buf["answer"][:, :self.P] = np.asarray(ans).reshape(self.N, self.P)
        self.rec[:, 0:2] = idim
        self.rec[:, 14:16] = adim

    def reset(self, mask=None):
        p = self._params()
        self._extras(p)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rmask = None if m is None else m.ctypes.data
        rc = emu_lib().emu_run(1, ctypes.byref(p))
        assert rc == 0, f"wave emulator reported error {rc}"

    def rollout(self, ingress, payload, op, flags=0):
        p = self._params()
        T = len(op)
        pay = np.ascontiguousarray(payload, np.int32)
        opa = np.ascontiguousarray(op, np.int32)
        reward = np.zeros((T, self.N), np.int32)
        term = np.zeros((T, self.N), np.uint8)
        p.sel, p.op, p.ingress, p.flags, p.n_steps = pay.ctypes.data, opa.ctypes.data, self.INGRESS[ingress], flags, T
        p.reward, p.term = reward.ctypes.data, term.ctypes.data
        rc = emu_lib().emu_run(3, ctypes.byref(p))
        assert rc == 0, f"wave emulator reported error {rc}"
        return reward, term

    def set_task_table(self, inputs, answers):
        T = len(inputs)
        self.tbl = [np.zeros((T, self.PS), np.int8), np.zeros((T, 2), np.int8), np.zeros((T, self.PS), np.int8),
                    np.zeros((T, 2), np.int8)]
        for j, (a, b) in enumerate(zip(inputs, answers)):
            self.tbl[0][j, :self.P].reshape(self.H, self.W)[:a.shape[0], :a.shape[1]] = a
            self.tbl[2][j, :self.P].reshape(self.H, self.W)[:b.shape[0], :b.shape[1]] = b
            self.tbl[1][j], self.tbl[3][j] = a.shape, b.shape

    def reset_from_table(self, idx, mask=None, aug_k=None, aug_perm=None):
        p = self._params()
        self._extras(p)
        idx = np.ascontiguousarray(idx, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rmask = None if m is None else m.ctypes.data
        p.task_idx = idx.ctypes.data
        if aug_k is not None:
            k8 = np.ascontiguousarray(aug_k, np.uint8)
            p.aug_k = k8.ctypes.data
        if aug_perm is not None:
            pm = np.zeros((self.N, 16), np.uint8)
            pm[:, :10] = aug_perm
            p.aug_perm = pm.ctypes.data
        rc = emu_lib().emu_run(2, ctypes.byref(p))
        assert rc == 0, f"wave emulator reported error {rc}"

    # ---- round-2 features (same kernels, the emulator runs them lock-step) ------------------------------
    def _extras(self, p):
        """Optional per-step outputs / sampler state shared by the step and reset kernels."""
        for name in ("trunc", "dense", "dense_cache", "episode", "cur_task"):
            arr = getattr(self, name, None)
            if arr is not None:
                setattr(p, name, arr.ctypes.data)
        p.step_limit = getattr(self, "step_limit", 0)
        if getattr(self, "_sampler", None):
            off, cnt, seed, base, aug = self._sampler
            p.pair_off, p.pair_cnt, p.n_problems = off.ctypes.data, cnt.ctypes.data, len(cnt)
            p.seed, p.env_base, p.aug_flags = seed, base, aug
        if getattr(self, "tbl", None) is not None:
            p.tbl_in, p.tbl_in_dim, p.tbl_ans, p.tbl_ans_dim = [t.ctypes.data for t in self.tbl]
            p.n_tasks = len(self.tbl[0])
        if getattr(self, "_pack", None) is not None:  # destination of STEP_PACK_OBS
            p.pack_out = self._pack.ctypes.data
        if getattr(self, "_flat", None) is not None:  # destination of STEP_FLAT_OBS
            out, L, filtered = self._flat
            p.flat_out, p.flat_stride, p.flat_filter = out.ctypes.data, out.shape[1], int(filtered)
            p.flat_tail = int(getattr(self, "_flat_tail", False))

    def set_packed_output(self):
        self._pack = np.full((self.N, (self.P + 7 + 15) & ~15), 0x55, np.uint8)

    def fused_packed(self):
        return self._pack.copy()

    def _flat_len(self, filtered):
        o2, clip = "selected" in self.buf, "clip" in self.buf
        return 3 * self.P + 10 if filtered else 2 * self.P + 6 + (self.P + 2 if clip else 0) + (4 * self.P + 6 if o2 else 0)

    def set_flat_output(self, filtered=False, tail=False):
        L = self._flat_len(filtered)
        self._flat = (np.full((self.N, ((L + 15) & ~15) + (16 if tail else 0)), 0x55, np.int8), L, filtered)
        self._flat_tail = tail

    def fused_flat(self):
        out, L, _ = self._flat
        assert not out[:, L:((L + 15) & ~15)].any()
        return out[:, :L].copy()

    def fused_tail(self):
        """int32 [N,4] view of the rows' tails: reward, action_steps, submit_count, terminated | truncated << 8 | status << 16."""
        out = self._flat[0]
        return np.ascontiguousarray(out[:, out.shape[1] - 16:]).view(np.int32).copy()

    def pack_mask_bits(self, masks):
        p = self._params()
        pay = np.ascontiguousarray(np.asarray(masks).astype(np.int8)).reshape(self.N, self.P)
        out = np.full((self.N, BITS_STRIDE), 0x55, np.uint8)
        p.sel, p.flat_out = pay.ctypes.data, out.ctypes.data
        rc = emu_lib().emu_run(8, ctypes.byref(p))
        assert rc == 0
        return out

    def set_state_rows(self, rows, mask=None):
        p = self._params()
        self._extras(p)
        rows = np.ascontiguousarray(rows, np.int8)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rows_in, p.rows_in_stride = rows.ctypes.data, rows.shape[1]
        p.rmask = None if m is None else m.ctypes.data
        rc = emu_lib().emu_run(6, ctypes.byref(p))
        assert rc == 0

    def transition_rows(self, rows, ingress, payload, op, src_env=None, tail=False, flags=0, in_place=False):
        p = self._params()
        self._extras(p)
        rows = np.ascontiguousarray(rows, np.int8)
        M = rows.shape[0]
        L = self._flat_len(False)
        out = np.full((M, ((L + 15) & ~15) + (16 if tail else 0)), 0x55, np.int8)
        if in_place:  # rows_out IS rows_in: untouched planes stay where they are (the library sets the writer's incremental mode)
            out[:, :L] = rows[:, :L]
            out[:, L:(L + 15) & ~15] = 0
            rows = out
            flags |= 512
        pay = (np.ascontiguousarray(np.asarray(payload).astype(np.int8)).reshape(M, self.P) if ingress == "mask"
               else np.ascontiguousarray(payload, np.int32))
        opa = np.ascontiguousarray(op, np.int32)
        reward, term = np.zeros(M, np.int32), np.zeros(M, np.uint8)
        src = None if src_env is None else np.ascontiguousarray(src_env, np.int32)
        p.n_resident, p.n_envs = self.N, M
        p.rows_in, p.rows_in_stride = rows.ctypes.data, rows.shape[1]
        p.flat_out, p.flat_stride, p.flat_filter, p.flat_tail = out.ctypes.data, out.shape[1], 0, int(tail)
        p.sel, p.op, p.ingress, p.flags = pay.ctypes.data, opa.ctypes.data, self.INGRESS[ingress], flags
        p.reward, p.term = reward.ctypes.data, term.ctypes.data
        p.task_idx = None if src is None else src.ctypes.data
        rc = emu_lib().emu_run(7, ctypes.byref(p))
        assert rc == 0, f"wave emulator reported error {rc}"
        return out, reward, term

    def set_truncation(self, limit):
        self.trunc, self.step_limit = np.zeros(self.N, np.uint8), int(limit)

    def set_dense_output(self):
        self.dense = np.zeros((self.N, 2), np.int32)
        self.dense_cache = np.zeros((self.N, 2), np.int32)  # (the library owns this one: arcle_set_dense_output allocates it)

    def set_sampler(self, pair_off, pair_cnt, seed, env_base=0, aug_flags=0):
        self._sampler = (np.ascontiguousarray(pair_off, np.int32), np.ascontiguousarray(pair_cnt, np.int32), int(seed), int(env_base), int(aug_flags))
        if getattr(self, "episode", None) is None:
            self.episode, self.cur_task = np.zeros(self.N, np.int32), np.full(self.N, -1, np.int32)

    def reset_sampled(self, mask=None):
        p = self._params()
        self._extras(p)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rmask = None if m is None else m.ctypes.data
        rc = emu_lib().emu_run(2, ctypes.byref(p))
        assert rc == 0, f"wave emulator reported error {rc}"

    def flat_obs(self, filtered=False):
        p = self._params()
        o2, clip = "selected" in self.buf, "clip" in self.buf
        L = 3 * self.P + 10 if filtered else 2 * self.P + 6 + (self.P + 2 if clip else 0) + (4 * self.P + 6 if o2 else 0)
        out = np.full((self.N, (L + 15) & ~15), 0x55, np.int8)
        p.flat_out, p.flat_stride, p.flat_filter = out.ctypes.data, out.shape[1], int(filtered)
        rc = emu_lib().emu_run(4, ctypes.byref(p))
        assert rc == 0 and not out[:, L:].any()
        return out[:, :L].copy()

    def packed_obs(self):
        p = self._params()
        R = (self.P + 7 + 15) & ~15
        out = np.full((self.N, R), 0x55, np.uint8)
        p.flat_out, p.flat_stride = out.ctypes.data, R
        rc = emu_lib().emu_run(5, ctypes.byref(p))
        assert rc == 0
        return out

    def step(self, ingress, payload, op, flags=0):
        p = self._params()
        self._extras(p)
        if ingress == "mask":
            pay = np.ascontiguousarray(np.asarray(payload).astype(np.int8)).reshape(self.N, self.P)
        elif ingress == "bits":
            pay = np.ascontiguousarray(payload, np.uint8).reshape(self.N, BITS_STRIDE)
        else:
            pay = np.ascontiguousarray(payload, np.int32)
        opa = np.ascontiguousarray(op if op is not None else np.zeros(self.N), np.int32)
        p.sel, p.op, p.ingress, p.flags = pay.ctypes.data, opa.ctypes.data, self.INGRESS[ingress], flags
        if ingress == "bbox5":
            p.op = None
        if getattr(self, "dense_cache", None) is not None and not flags & 16:
            self.dense_cache[:] = 0  # (the launcher's rule, arcle_hip.hip launch_step: a step without ARCLE_STEP_DENSE drops the pair cache)
        rc = emu_lib().emu_run(0, ctypes.byref(p))
        assert rc == 0, f"wave emulator reported error {rc} (divergent cross-lane op / non-uniform value)"
        return self.reward.copy(), self.term.copy()

    def get(self, field):
        if field in self.buf:
            return self.plane(field).copy()
        off, n = REC[field]
        return self.rec[:, off:off + n].copy()

    def padding_is_zero(self):
        return all(not b[:, self.P:].any() for b in self.buf.values())

    def counters(self):
        return self.cnt.copy()

    def status(self, clear=True):
        s = int(self.stat[0])
        if clear:
            self.stat[0] = 0
        return s


# ---- emulator of the workgroup-per-env kernels for grids beyond 1024 cells (tests/emu/big_emu.cpp runs arcle_big.h on host threads) ----
class _BigParams(ctypes.Structure):  # mirror of arcle_big::BigParams (arcle_amd/csrc/arcle_big.h)
    _fields_ = [("plane", ctypes.c_void_p * 8), ("rec", ctypes.c_void_p), ("cnt", ctypes.c_void_p),
                ("op", ctypes.c_void_p), ("sel", ctypes.c_void_p), ("reward", ctypes.c_void_p), ("term", ctypes.c_void_p),
                ("n_envs", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("P", ctypes.c_int32),
                ("PS", ctypes.c_int32), ("n_ops", ctypes.c_int32), ("max_trial", ctypes.c_int32), ("ingress", ctypes.c_int32),
                ("flags", ctypes.c_uint32), ("step_limit", ctypes.c_int32), ("status", ctypes.c_void_p),
                ("d_ops", ctypes.c_void_p), ("trunc", ctypes.c_void_p), ("flat_out", ctypes.c_void_p),
                ("flat_stride", ctypes.c_int32), ("flat_filter", ctypes.c_int32), ("flat_tail", ctypes.c_int32),
                ("flat_seq", ctypes.c_int32), ("pack_out", ctypes.c_void_p), ("rmask", ctypes.c_void_p),
                ("task_idx", ctypes.c_void_p), ("tbl_in", ctypes.c_void_p), ("tbl_ans", ctypes.c_void_p),
                ("tbl_in_dim", ctypes.c_void_p), ("tbl_ans_dim", ctypes.c_void_p), ("n_tasks", ctypes.c_int32),
                ("seed", ctypes.c_uint64), ("env_base", ctypes.c_int64), ("episode", ctypes.c_void_p),
                ("cur_task", ctypes.c_void_p), ("pair_off", ctypes.c_void_p), ("pair_cnt", ctypes.c_void_p),
                ("n_problems", ctypes.c_int32), ("rows_in", ctypes.c_void_p), ("rows_in_stride", ctypes.c_int32),
                ("n_resident", ctypes.c_int32), ("src_env", ctypes.c_void_p), ("res_answer", ctypes.c_void_p), ("res_rec", ctypes.c_void_p),
                ("aug_flags", ctypes.c_uint32), ("aug_k", ctypes.c_void_p), ("aug_perm", ctypes.c_void_p), ("acct", ctypes.c_void_p), ("dense", ctypes.c_void_p),
                ("w_magic", ctypes.c_uint32)]


_big_emu = None


def big_emu_lib():
    global _big_emu
    if _big_emu is None:
        d = os.path.join(ROOT, "tests", "emu")
        so, src = os.path.join(d, "libbig_emu.so"), os.path.join(d, "big_emu.cpp")
        hdr = os.path.join(ROOT, "arcle_amd", "csrc", "arcle_big.h")
        if os.environ.get("ARCLE_BIG_EMU_LIB"):  # (tests/test_emu_sanitized.py: the ASan + UBSan build)
            so = os.environ["ARCLE_BIG_EMU_LIB"]
        elif not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-o", so, src])
        _big_emu = ctypes.CDLL(so)
        _big_emu.big_emu_run.argtypes = [ctypes.c_int, ctypes.POINTER(_BigParams), ctypes.c_int, ctypes.c_int]
        assert _big_emu.big_emu_params_size() == ctypes.sizeof(_BigParams), "BigParams layout drifted"
    return _big_emu


class BigEmuBackend(EmuBackend):
    """arcle_big.h (one workgroup per env; H * W > 1024) on host threads.  Same surface as EmuBackend where the big path has the feature."""
    name = "bigemu"
    THREADS = 16  # the emulated workgroup (the product launches 128-512: every loop of the body is strided by the thread count)
    LEAN, CPT = True, 0  # CPT: the LEAN instantiation's compile-time bound on the chunks per thread (0: the run-time loop, any plane on 16 threads)
    LEAN_FLAGS = 1 | 2 | 4 | 8 | 64  # arcle_big.h LEAN_FLAGS: AUTORESET | ELIDE_SELECTED | TRUNCATE | RESAMPLE | RESET_ON_SUBMIT

    def _params(self):
        p = _BigParams()
        for i, k in enumerate(PLANES):
            p.plane[i] = self.buf[k].ctypes.data if k in self.buf else None
        p.rec, p.cnt = self.rec.ctypes.data, self.cnt.ctypes.data
        p.reward, p.term, p.status = self.reward.ctypes.data, self.term.ctypes.data, self.stat.ctypes.data
        p.n_envs, p.H, p.W, p.P, p.PS = self.N, self.H, self.W, self.P, self.PS
        p.max_trial, p.n_ops = self.max_trial, len(self.ops)
        self._ops_arr = np.zeros(65, np.uint32)
        self._ops_arr[:len(self.ops)] = self.ops
        p.d_ops = self._ops_arr.ctypes.data
        if getattr(self, "count_bytes", False):
            p.acct = self.acct.ctypes.data  # uint32 [2][N]: bytes without the row padding / bytes issued
        return p

    def set_dense_output(self):
        self.dense = np.full((self.N, 2), -7, np.int32)

    def _extras(self, p):
        for name in ("trunc", "episode", "cur_task", "dense"):
            arr = getattr(self, name, None)
            if arr is not None:
                setattr(p, name, arr.ctypes.data)
        p.step_limit = getattr(self, "step_limit", 0)
        if getattr(self, "_sampler", None):
            off, cnt, seed, base, aug = self._sampler
            p.pair_off, p.pair_cnt, p.n_problems = off.ctypes.data, cnt.ctypes.data, len(cnt)
            p.seed, p.env_base, p.aug_flags = seed, base, aug
        if getattr(self, "tbl", None) is not None:
            p.tbl_in, p.tbl_in_dim, p.tbl_ans, p.tbl_ans_dim = [t.ctypes.data for t in self.tbl]
            p.n_tasks = len(self.tbl[0])
        if getattr(self, "_pack", None) is not None:
            p.pack_out = self._pack.ctypes.data
        if getattr(self, "_flat", None) is not None:
            out, L, filtered = self._flat
            p.flat_out, p.flat_stride, p.flat_filter = out.ctypes.data, out.shape[1], int(filtered)
            p.flat_tail = int(getattr(self, "_flat_tail", False))
            p.flat_seq = int(getattr(self, "_flat_seq", 0))

    def _run(self, what, p, mode=0):
        nch = self.PS // 16
        nthreads = max(self.THREADS, ((nch + self.CPT - 1) // self.CPT + 15) & ~15) if self.CPT else self.THREADS
        rc = big_emu_lib().big_emu_run(what, ctypes.byref(p), mode, nthreads)
        assert rc == 0, f"big-grid emulator reported error {rc}"

    def reset(self, mask=None):
        p = self._params()
        self._extras(p)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rmask = None if m is None else m.ctypes.data
        self._run(1, p, 0)

    def reset_from_table(self, idx, mask=None, aug_k=None, aug_perm=None):
        p = self._params()
        self._extras(p)
        idx = np.ascontiguousarray(idx, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rmask = None if m is None else m.ctypes.data
        p.task_idx = idx.ctypes.data
        if aug_k is not None:
            k8 = np.ascontiguousarray(aug_k, np.uint8)
            p.aug_k = k8.ctypes.data
        if aug_perm is not None:
            pm = np.zeros((self.N, 16), np.uint8)
            pm[:, :10] = aug_perm
            p.aug_perm = pm.ctypes.data
        self._run(1, p, 1)

    def set_sampler(self, pair_off, pair_cnt, seed, env_base=0, aug_flags=0):
        self._sampler = (np.ascontiguousarray(pair_off, np.int32), np.ascontiguousarray(pair_cnt, np.int32), seed, env_base, aug_flags)
        self.episode = np.zeros(self.N, np.int32)
        self.cur_task = np.full(self.N, -1, np.int32)

    def reset_sampled(self, mask=None):
        p = self._params()
        self._extras(p)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rmask = None if m is None else m.ctypes.data
        self._run(1, p, 2)

    def set_truncation(self, limit):
        self.trunc = np.zeros(self.N, np.uint8)
        self.step_limit = int(limit)

    def step(self, ingress, payload, op, flags=0):
        p = self._params()
        self._extras(p)
        if ingress == "mask":
            pay = np.ascontiguousarray(np.asarray(payload).astype(np.int8)).reshape(self.N, self.P)
        elif ingress == "bits":
            pay = np.ascontiguousarray(payload, np.uint8).reshape(self.N, self.PS // 8)
        else:
            pay = np.ascontiguousarray(payload, np.int32)
        opa = np.ascontiguousarray(op if op is not None else np.zeros(self.N), np.int32)
        p.sel, p.op, p.ingress, p.flags = pay.ctypes.data, opa.ctypes.data, self.INGRESS[ingress], flags
        # the instantiation the product's launcher would pick (arcle_big.hip launch_step): LEAN when the flag set, the plane width and
        # the launch allow it — with the product's one / two / four chunks per thread in the BigEmuOne / Two / FourBackend —, the generic body otherwise
        lean = self.LEAN and not (flags & ~self.LEAN_FLAGS) and self.W >= 16 and not getattr(self, "count_bytes", False)
        self.lean_steps = getattr(self, "lean_steps", 0) + int(bool(lean))
        if lean:
            self._run(4, p, self.CPT)
        else:
            self._run(0, p)
        return self.reward.copy(), self.term.copy()

    def pack_mask_bits(self, masks):
        p = self._params()
        pay = np.ascontiguousarray(np.asarray(masks).astype(np.int8)).reshape(self.N, self.P)
        out = np.full((self.N, self.PS // 8), 0x55, np.uint8)
        p.sel, p.pack_out = pay.ctypes.data, out.ctypes.data
        self._run(2, p, 2)
        return out

    def flat_obs(self, filtered=False):
        L = self._flat_len(filtered)
        out = np.full((self.N, (L + 15) & ~15), 0x55, np.int8)
        p = self._params()
        p.flat_out, p.flat_stride, p.flat_filter = out.ctypes.data, out.shape[1], int(filtered)
        self._run(2, p, 0)
        assert not out[:, L:].any()
        return out[:, :L].copy()

    def packed_obs(self):
        out = np.full((self.N, (self.P + 7 + 15) & ~15), 0x55, np.uint8)
        p = self._params()
        p.pack_out = out.ctypes.data
        self._run(2, p, 1)
        return out

    def set_state_rows(self, rows, mask=None):
        p = self._params()
        rows = np.ascontiguousarray(rows, np.int8)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        p.rows_in, p.rows_in_stride = rows.ctypes.data, rows.shape[1]
        p.rmask = None if m is None else m.ctypes.data
        self._run(3, p)

    def transition_rows(self, rows, ingress, payload, op, src_env=None, tail=False, flags=0, in_place=False):
        """What arcle_transition_rows does for a big-grid handle (arcle_hip.hip): rows -> scratch envs (+ the answer of resident env
        src_env[r]), one step of the scratch envs with the fused row writer."""
        rows = np.ascontiguousarray(rows, np.int8)
        M, L = rows.shape[0], self._flat_len(False)
        out = np.full((M, ((L + 15) & ~15) + (16 if tail else 0)), 0x55, np.int8)
        if in_place:
            out[:, :L] = rows[:, :L]
            rows = out
        scratch = {k: np.full((M, self.PS), 0x33, np.int8) for k in self.buf}
        rec, cnt = np.full((M, 16), 0x33, np.int8), np.full((M, 2), 0x33, np.int32)
        reward, term = np.zeros(M, np.int32), np.zeros(M, np.uint8)
        p = self._params()
        for i, k in enumerate(PLANES):
            p.plane[i] = scratch[k].ctypes.data if k in scratch else None
        p.rec, p.cnt, p.n_envs = rec.ctypes.data, cnt.ctypes.data, M
        p.reward, p.term = reward.ctypes.data, term.ctypes.data
        p.n_resident, p.res_answer, p.res_rec = self.N, self.buf["answer"].ctypes.data, self.rec.ctypes.data
        src = None if src_env is None else np.ascontiguousarray(src_env, np.int32)
        p.src_env = None if src is None else src.ctypes.data
        p.rows_in, p.rows_in_stride = rows.ctypes.data, rows.shape[1]
        self._run(3, p)
        pay = (np.ascontiguousarray(np.asarray(payload).astype(np.int8)).reshape(M, self.P) if ingress == "mask"
               else np.ascontiguousarray(payload, np.int32))
        opa = np.ascontiguousarray(op, np.int32)
        p.sel, p.op, p.ingress, p.flags = pay.ctypes.data, opa.ctypes.data, self.INGRESS[ingress], flags | 128
        p.flat_out, p.flat_stride, p.flat_filter, p.flat_tail = out.ctypes.data, out.shape[1], 0, int(tail)
        self._run(0, p)
        return out, reward, term


class BigEmuGenericBackend(BigEmuBackend):
    """... every step through the generic instantiation (what tuning launches, accounting and transition_rows run)."""
    name = "bigemu_generic"
    LEAN = False


class BigEmuOneBackend(BigEmuBackend):
    """... with at least one host thread per plane chunk: the one-chunk-per-thread LEAN instantiation (every plane of up to 8192 cells on
    the GPU).  Hundreds of host threads behind a pthread barrier: small cases only."""
    name = "bigemu_one"
    CPT = 1


class BigEmuTwoBackend(BigEmuBackend):
    """... with one host thread per TWO plane chunks: the instantiation the product's launcher picks by default (64 threads at 40 x 40,
    128 at 64 x 64, 512 at 127 x 127)."""
    name = "bigemu_two"
    CPT = 2


class BigEmuFourBackend(BigEmuBackend):
    """... four chunks per thread: the template's general form (the product launches one or two, arcle_big.hip launch_step: four lost on
    MI355X, profiles/round6_experiments.txt §2d)."""
    name = "bigemu_four"
    CPT = 4


# ---- HIP (the product, through arcle_amd.engine -> libarcle_hip.so C ABI) ------------------------------
class HipBackend:
    name = "hip"

    def __init__(self, N, H, W, max_trial, kind, ops):
        import torch
        from arcle_amd.engine import EnvBatch
        self.torch = torch
        self.N, self.H, self.W, self.kind = N, H, W, kind
        self.b = EnvBatch(N, H, W, max_trial, kind)
        self.b.set_op_table(ops)

    def set_tasks(self, inp, idim, ans, adim):
        self.b.set_tasks_padded(inp, idim, ans, adim)

    def reset(self, mask=None):
        self.b.reset(mask)

    def step(self, ingress, payload, op, flags=0):
        t = self.torch
        dev = self.b.device
        if ingress == "bbox5":
            r, tm = self.b.step_bbox5(t.as_tensor(np.ascontiguousarray(payload, np.int32), device=dev), flags)
            return r.cpu().numpy().copy(), tm.cpu().numpy().copy()
        opt = t.as_tensor(np.ascontiguousarray(op, np.int32), device=dev)
        if ingress == "bits":
            r, tm = self.b.step_bits(t.as_tensor(np.ascontiguousarray(payload, np.uint8), device=dev), opt, flags)
            return r.cpu().numpy().copy(), tm.cpu().numpy().copy()
        if ingress == "mask":
            pay = t.as_tensor(np.ascontiguousarray(np.asarray(payload).astype(np.int8)), device=dev).reshape(self.N, self.H, self.W)
            r, tm = self.b.step_mask(pay, opt, flags)
        else:
            pay = t.as_tensor(np.ascontiguousarray(payload, np.int32), device=dev)
            r, tm = (self.b.step_bbox if ingress == "bbox" else self.b.step_point)(pay, opt, flags)
        return r.cpu().numpy().copy(), tm.cpu().numpy().copy()

    def get(self, field):
        if field in self.b.planes:
            return self.b.plane(field).cpu().numpy().copy()
        return self.b.field(field).cpu().numpy().copy()

    def rollout(self, ingress, payload, op, flags=0):
        t = self.torch
        r, tm = self.b.rollout(t.as_tensor(np.ascontiguousarray(payload, np.int32), device=self.b.device),
                               t.as_tensor(np.ascontiguousarray(op, np.int32), device=self.b.device), flags,
                               point=(ingress == "point"))
        return r.cpu().numpy(), tm.cpu().numpy()

    def set_task_table(self, inputs, answers):
        self.b.set_task_table(inputs, answers)

    def reset_from_table(self, idx, mask=None, aug_k=None, aug_perm=None):
        t = self.torch
        self.b.reset_from_table(t.as_tensor(np.ascontiguousarray(idx, np.int32), device=self.b.device),
                                None if mask is None else t.as_tensor(np.ascontiguousarray(mask, np.uint8), device=self.b.device),
                                None if aug_k is None else t.as_tensor(np.ascontiguousarray(aug_k, np.uint8)),
                                None if aug_perm is None else t.as_tensor(np.ascontiguousarray(aug_perm, np.uint8)))

    def padding_is_zero(self):
        return all(not bool(p[:, self.b.P:].any()) for p in self.b.planes.values())

    # ---- round-2 features -------------------------------------------------------------------------------
    def set_truncation(self, limit):
        self.b.set_truncation(limit)

    def set_dense_output(self):
        self.b.set_dense_output()

    def set_sampler(self, pair_off, pair_cnt, seed, env_base=0, aug_flags=0):
        self.b.set_sampler(pair_off, pair_cnt, seed, env_base, aug_flags)

    def reset_sampled(self, mask=None):
        self.b.reset_sampled(None if mask is None else self.torch.as_tensor(np.ascontiguousarray(mask, np.uint8), device=self.b.device))

    def flat_obs(self, filtered=False):
        return self.b.flat_obs(filtered=filtered).cpu().numpy().copy()

    def set_flat_output(self, filtered=False, tail=False):
        self.b.set_flat_output(filtered, tail)
        self.b._flat_buf.fill_(0x55)

    def fused_flat(self):
        self.torch.cuda.synchronize()
        L = self.b.flat.shape[1]
        assert not self.b._flat_buf[:, L:(L + 15) & ~15].any()
        return self.b.flat.cpu().numpy().copy()

    def fused_tail(self):
        self.torch.cuda.synchronize()
        return self.b.flat_tail.cpu().numpy().copy()

    def pack_mask_bits(self, masks):
        t = self.torch
        return self.b.pack_mask_bits(t.as_tensor(np.ascontiguousarray(np.asarray(masks).astype(np.int8)), device=self.b.device)).cpu().numpy()

    def set_state_rows(self, rows, mask=None):
        t = self.torch
        self.b.set_state_rows(t.as_tensor(np.ascontiguousarray(rows, np.int8), device=self.b.device),
                              None if mask is None else t.as_tensor(np.ascontiguousarray(mask, np.uint8), device=self.b.device))

    def transition_rows(self, rows, ingress, payload, op, src_env=None, tail=False, flags=0, in_place=False):
        t, dev = self.torch, self.b.device
        M = len(rows)
        if in_place:
            L = self.b.state_row_size()
            buf = t.zeros((M, ((L + 15) & ~15) + (16 if tail else 0)), dtype=t.int8, device=dev)
            buf[:, :L] = t.as_tensor(np.ascontiguousarray(rows, np.int8), device=dev)[:, :L]
            pay = (t.as_tensor(np.ascontiguousarray(np.asarray(payload).astype(np.int8)), device=dev).reshape(M, self.H, self.W) if ingress == "mask"
                   else t.as_tensor(np.ascontiguousarray(payload, np.int32), device=dev))
            out, r, tm = self.b.transition_rows(buf, ingress, pay, t.as_tensor(np.ascontiguousarray(op, np.int32), device=dev),
                                                None if src_env is None else t.as_tensor(np.ascontiguousarray(src_env, np.int32), device=dev),
                                                out=buf, tail=tail, flags=flags)
            return out.cpu().numpy(), r.cpu().numpy(), tm.cpu().numpy()
        pay = (t.as_tensor(np.ascontiguousarray(np.asarray(payload).astype(np.int8)), device=dev).reshape(M, self.H, self.W) if ingress == "mask"
               else t.as_tensor(np.ascontiguousarray(payload, np.int32), device=dev))
        out, r, tm = self.b.transition_rows(t.as_tensor(np.ascontiguousarray(rows, np.int8), device=dev), ingress, pay,
                                            t.as_tensor(np.ascontiguousarray(op, np.int32), device=dev),
                                            None if src_env is None else t.as_tensor(np.ascontiguousarray(src_env, np.int32), device=dev),
                                            tail=tail, flags=flags)
        return out.cpu().numpy(), r.cpu().numpy(), tm.cpu().numpy()

    def packed_obs(self):
        return self.b.packed_obs().cpu().numpy().copy()

    def set_packed_output(self):
        self.b.set_packed_output().fill_(0x55)

    def fused_packed(self):
        self.torch.cuda.synchronize()
        return self.b.packed.cpu().numpy().copy()

    @property
    def trunc(self):
        return self.b.trunc.cpu().numpy()

    @property
    def dense(self):
        return self.b.dense.cpu().numpy()

    @property
    def episode(self):
        return self.b.episode.cpu().numpy()

    @property
    def cur_task(self):
        return self.b.cur_task.cpu().numpy()

    def counters(self):
        return self.b.cnt.cpu().numpy().copy()

    def status(self, clear=True):
        return self.b.status(clear)


BACKENDS = {"oracle": OracleBackend, "emu": EmuBackend, "hip": HipBackend}


# ---- golden fixtures ------------------------------------------------------------------------------------
def fixture_names():
    """Trace fixtures of tests/golden/make_golden.py (research.npz has its own layout, tests/features.py)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f not in ("research.npz", "api_edge.npz") and not f.startswith("big_"))


def big_fixture_names():
    """Trace fixtures of tests/golden/make_golden_big.py: max_grid_size beyond 1024 cells (the workgroup-per-env kernels)."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("big_"))


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    fx = {k: z[k] for k in z.files}
    fx["meta"] = json.loads(str(fx["meta"]))
    fx["fields"] = json.loads(str(fx["fields"]))
    return fx


STEP_ELIDE_SELECTED = 2


def can_elide(ops):
    """ARCLE_STEP_ELIDE_SELECTED is only valid for tables without keep_sel-wrapped ops (include/arcle_hip.h)."""
    return not any((d >> 16) & O.F_KEEP_SEL for d in ops)


def replay_fixture(backend_cls, name, max_steps=None, flags=0):
    """Replays a golden trace set on a backend; returns a list of mismatch descriptions (empty = parity)."""
    fx = load_fixture(name)
    m = fx["meta"]
    N, S = m["N"], m["S"] if max_steps is None else min(m["S"], max_steps)
    be = backend_cls(N, m["H"], m["W"], m["max_trial"], m["kind"], m["ops"])
    be.set_tasks(fx["input"], fx["input_dim"], fx["answer"], fx["answer_dim"])
    be.reset()
    mask_idx = {int(s): i for i, s in enumerate(fx["mask_steps"])}
    full_idx = {int(s): i for i, s in enumerate(fx["full_steps"])}
    errs = []
    for s in range(S):
        ing = int(fx["ingress"][s])
        if ing == 0:
            r, t = be.step("bbox", fx["bbox"][s], fx["op"][s], flags)
        elif ing == 1:
            r, t = be.step("point", fx["xy"][s], fx["op"][s], flags)
        else:
            r, t = be.step("mask", fx["masks"][mask_idx[s]], fx["op"][s], flags)
        if not np.array_equal(r, fx["reward"][s]):
            errs.append(f"{name} step {s}: reward {r.tolist()} != {fx['reward'][s].tolist()}")
        if not np.array_equal(t, fx["term"][s]):
            errs.append(f"{name} step {s}: terminated {t.tolist()} != {fx['term'][s].tolist()}")
        cnt = be.counters()
        if not np.array_equal(cnt[:, 0], fx["steps"][s]):
            errs.append(f"{name} step {s}: steps counter mismatch")
        if m["kind"] != "raw" and not np.array_equal(cnt[:, 1], fx["submit_count"][s]):
            errs.append(f"{name} step {s}: submit_count mismatch")
        for fi, f in enumerate(fx["fields"]):
            got = be.get(f)
            h = checksum(got)
            bad = np.nonzero(h != fx["hash"][s, :, fi])[0]
            if bad.size:
                errs.append(f"{name} step {s} field {f}: checksum mismatch for envs {bad.tolist()} "
                            f"(ops {fx['op'][s][bad].tolist()}, ingress {ing})")
            if s in full_idx and not np.array_equal(got, fx["full_" + f][full_idx[s]]):
                errs.append(f"{name} step {s} field {f}: full state mismatch")
        if be.status():
            errs.append(f"{name} step {s}: unexpected device status flag")
        if len(errs) > 20:
            break
    if hasattr(be, "padding_is_zero") and not be.padding_is_zero():
        errs.append(f"{name}: plane padding bytes (cells >= H*W) are not zero")
    return errs


# ---- random differential traces (no reference needed: backend vs oracle) ---------------------------------
def random_trace_compare(backend_cls, kind, ops, H, W, N, S, seed, max_trial=-1, flags=0, op_weights=None,
                         bad_ops=False, new_forms=False, int8_masks=False):
    """Steps `backend_cls` and the oracle side by side on seeded random tasks/actions; returns mismatches.
    new_forms: the backend under test receives bbox actions as 5-tuple records (bbox5) and masks bit-packed (bits; the masks are
    boolean then) — the oracle gets the classic forms of the same actions.
    int8_masks: every action is a full mask of arbitrary int8 values (out of the Gym contract, but NumPy takes them: sel > 0 / != 0 / sum /
    argmax differ then): sparse values in [-3, 3], one cell of 1 / 2 / -1 / 127 / -128, the pair {2, -1} (sum 1, arg-max at the 2), 0 / 1
    noise with a few negatives, a block of ones holding one 5."""
    rng = np.random.default_rng(seed)
    be = backend_cls(N, H, W, max_trial, kind, ops)
    orc = OracleBackend(N, H, W, max_trial, kind, ops)
    inp = np.zeros((N, H, W), np.int8)
    ans = np.zeros((N, H, W), np.int8)
    idim = np.zeros((N, 2), np.int8)
    adim = np.zeros((N, 2), np.int8)
    for n in range(N):
        ih, iw = rng.integers(1, H + 1), rng.integers(1, W + 1)
        ncol = [10, 10, 2, 3][rng.integers(0, 4)]
        g = rng.integers(0, ncol, (ih, iw)).astype(np.int8) * (rng.random((ih, iw)) < [1.0, 0.5][rng.integers(0, 2)])
        inp[n, :ih, :iw] = g
        idim[n] = (ih, iw)
        if rng.random() < 0.5:
            ans[n, :ih, :iw] = g
            adim[n] = (ih, iw)
        else:
            ah, aw = rng.integers(1, H + 1), rng.integers(1, W + 1)
            ans[n, :ah, :aw] = rng.integers(0, 10, (ah, aw))
            adim[n] = (ah, aw)
    for b in (be, orc):
        b.set_tasks(inp, idim, ans, adim)
        b.reset()
    n_ops = len(ops)
    w = np.ones(n_ops) if op_weights is None else np.asarray(op_weights, float)
    w = w / w.sum()
    errs = []
    fields = [f for f in PLANES[:-1] if f in O.KIND_PLANES[kind]] + [
        f for f in REC if f != "answer_dim" and (kind == "o2arc" or f in ("input_dim", "grid_dim", "trials_remain", "terminated")
                                                  or (kind == "arc" and f == "clip_dim"))]
    for s in range(S):
        op = rng.choice(n_ops, size=N, p=w).astype(np.int32)
        if bad_ops and s % 7 == 3:
            op[rng.integers(0, N)] = n_ops + rng.integers(0, 3)
        ing = "mask" if int8_masks else ["bbox", "bbox", "point", "mask"][rng.integers(0, 4)]
        if int8_masks:
            pay = np.zeros((N, H, W), np.int8)
            for n in range(N):
                t = rng.integers(0, 6)
                x, y = rng.integers(0, H), rng.integers(0, W)
                if t == 1:
                    pay[n] = rng.integers(-3, 4, (H, W)) * (rng.random((H, W)) < rng.random() * 0.1)
                elif t == 2:
                    pay[n, x, y] = [1, 2, -1, 127, -128][rng.integers(0, 5)]
                elif t == 3:
                    pay[n, x, y] = 2
                    pay[n, rng.integers(0, H), rng.integers(0, W)] -= 1
                elif t == 4:
                    pay[n] = rng.random((H, W)) < rng.random() * 0.3
                    pay[n][rng.random((H, W)) < 0.01] = -1
                elif t == 5:
                    pay[n, x:x + rng.integers(1, 6), y:y + rng.integers(1, 6)] = 1
                    pay[n, x, y] = 5
        elif ing == "bbox":
            pay = np.stack([rng.integers(0, H, N), rng.integers(0, W, N), rng.integers(0, H, N), rng.integers(0, W, N)], 1)
            small = rng.random(N) < 0.5
            pay[small, 2] = np.minimum(H - 1, pay[small, 0] + rng.integers(0, 4, small.sum()))
            pay[small, 3] = np.minimum(W - 1, pay[small, 1] + rng.integers(0, 4, small.sum()))
        elif ing == "point":
            pay = np.stack([rng.integers(0, H, N), rng.integers(0, W, N)], 1)
        else:
            pay = np.zeros((N, H, W), np.int8)
            for n in range(N):
                t = rng.integers(0, 4)
                if t == 1:
                    pay[n] = rng.random((H, W)) < rng.random() * 0.3
                elif t == 2:
                    x, y = rng.integers(0, H), rng.integers(0, W)
                    pay[n, x, y] = 1
                elif t == 3:
                    x, y = rng.integers(0, H), rng.integers(0, W)
                    pay[n, x:x + rng.integers(1, 5), y:y + rng.integers(1, 5)] = 1
        if new_forms and ing == "bbox":
            r1, t1 = be.step("bbox5", np.concatenate([pay, op[:, None]], 1), None, flags)
        elif new_forms and ing == "mask":
            r1, t1 = be.step("bits", pack_bits(pay), op, flags)
        else:
            r1, t1 = be.step(ing, pay, op, flags)
        r2, t2 = orc.step(ing, pay, op, flags)
        tag = f"{kind} {H}x{W} seed {seed} step {s} ingress {ing}{' (new form)' if new_forms else ''}"
        if not np.array_equal(r1, r2):
            errs.append(f"{tag}: reward mismatch envs {np.nonzero(r1 != r2)[0].tolist()}")
        if not np.array_equal(t1, t2):
            errs.append(f"{tag}: terminated mismatch envs {np.nonzero(t1 != t2)[0].tolist()}")
        if not np.array_equal(be.counters(), orc.counters()):
            errs.append(f"{tag}: counters mismatch")
        s1, s2 = be.status(), orc.status()
        if s1 != s2:
            errs.append(f"{tag}: status {s1} vs oracle {s2}")
        for f in fields:
            a, b = be.get(f), orc.get(f)
            if not np.array_equal(a, b):
                bad = np.nonzero((a != b).reshape(N, -1).any(1))[0]
                errs.append(f"{tag} field {f}: envs {bad.tolist()} ops {op[bad].tolist()}")
        if len(errs) > 12:
            break
    return errs


def task_table_compare(backend_cls, H, W, N, T, seed):
    """reset_from_table (device task table, per-env task index, optional mask) vs the oracle's set_tasks + reset."""
    rng = np.random.default_rng(seed)
    ins = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
    outs = [rng.integers(0, 10, (rng.integers(1, H + 1), rng.integers(1, W + 1))).astype(np.int8) for _ in range(T)]
    ops = O.o2arc_ops()
    be = backend_cls(N, H, W, 5, "o2arc", ops)
    orc = OracleBackend(N, H, W, 5, "o2arc", ops)
    be.set_task_table(ins, outs)
    idx = rng.integers(0, T, N)
    be.reset_from_table(idx)
    orc.env.set_tasks([ins[i] for i in idx], [outs[i] for i in idx])
    orc.reset()
    errs = []

    def check(tag):
        for f in PLANES + list(REC):
            if not np.array_equal(be.get(f), orc.get(f)):
                errs.append(f"{tag}: field {f} differs")
        if not np.array_equal(be.counters(), orc.counters()):
            errs.append(f"{tag}: counters differ")

    check("after reset_from_table")
    for s in range(12):  # dirty the state, then reset a masked subset onto new tasks
        op = rng.integers(0, 35, N).astype(np.int32)
        bb = np.stack([rng.integers(0, H, N), rng.integers(0, W, N), rng.integers(0, H, N), rng.integers(0, W, N)], 1)
        be.step("bbox", bb, op)
        orc.step("bbox", bb, op)
    if be.status() != orc.status():  # (non-square grids: Rotate may flag ARCLE_ST_ROTATE_DOMAIN on both sides)
        errs.append("status flags differ after the random steps")
    mask = (rng.random(N) < 0.5).astype(np.uint8)
    idx2 = rng.integers(0, T, N)
    be.reset_from_table(idx2, mask)
    sel = np.nonzero(mask)[0]
    new_in = [ins[idx2[n]] if mask[n] else ins[idx[n]] for n in range(N)]
    new_out = [outs[idx2[n]] if mask[n] else outs[idx[n]] for n in range(N)]
    orc.env.set_tasks(new_in, new_out)
    orc.reset(mask)
    check("after masked reset_from_table")
    bad = idx2.copy()
    bad[0] = T + 3
    be.reset_from_table(bad, np.ones(N, np.uint8))
    if not be.status() & 4:
        errs.append("out-of-range task index did not raise ARCLE_ST_BAD_TASK")
    return errs


def rollout_compare(backend_cls, kind, ops, H, W, N, T, seed, ingress="bbox", flags=0, max_trial=3):
    """One T-step rollout launch vs T sequential oracle steps: per-step reward/terminated, final state, counters."""
    rng = np.random.default_rng(seed)
    be = backend_cls(N, H, W, max_trial, kind, ops)
    orc = OracleBackend(N, H, W, max_trial, kind, ops)
    inp = np.zeros((N, H, W), np.int8)
    idim = np.zeros((N, 2), np.int8)
    for n in range(N):
        ih, iw = rng.integers(1, H + 1), rng.integers(1, W + 1)
        inp[n, :ih, :iw] = rng.integers(0, 4, (ih, iw))
        idim[n] = (ih, iw)
    for b in (be, orc):
        b.set_tasks(inp, idim, inp.copy(), idim.copy())
        b.reset()
    op = rng.integers(0, len(ops), (T, N)).astype(np.int32)
    op[rng.random((T, N)) < 0.05] = len(ops) - 1  # some submits (answer == input: terminations, trial counting)
    if ingress == "bbox":
        pay = np.stack([rng.integers(0, H, (T, N)), rng.integers(0, W, (T, N)), rng.integers(0, H, (T, N)),
                        rng.integers(0, W, (T, N))], -1).astype(np.int32)
        small = rng.random((T, N)) < 0.5
        pay[..., 2] = np.where(small, np.minimum(H - 1, pay[..., 0] + rng.integers(0, 3, (T, N))), pay[..., 2])
        pay[..., 3] = np.where(small, np.minimum(W - 1, pay[..., 1] + rng.integers(0, 3, (T, N))), pay[..., 3])
    else:
        pay = np.stack([rng.integers(0, H, (T, N)), rng.integers(0, W, (T, N))], -1).astype(np.int32)
    r1, t1 = be.rollout(ingress, pay, op, flags)
    errs = []
    for t in range(T):
        r2, t2 = orc.step(ingress, pay[t], op[t], flags)
        if not np.array_equal(r1[t], r2):
            errs.append(f"step {t}: reward differs for envs {np.nonzero(r1[t] != r2)[0].tolist()}")
        if not np.array_equal(t1[t], t2):
            errs.append(f"step {t}: terminated differs for envs {np.nonzero(t1[t] != t2)[0].tolist()}")
    for f in [f for f in PLANES if f in O.KIND_PLANES[kind]] + list(REC):
        if f in ("clip_dim", "object_dim", "object_pos", "active", "rotation_parity") and kind != "o2arc" and not (kind == "arc" and f == "clip_dim"):
            continue
        if not np.array_equal(be.get(f), orc.get(f)):
            errs.append(f"final state: field {f} differs")
    if not np.array_equal(be.counters(), orc.counters()):
        errs.append("final counters differ")
    if be.status() != orc.status():
        errs.append("status flags differ")
    return errs


def spiral_grid(H, W):
    """A 1-cell-wide spiral corridor of colour 1 in a field of colour 2 (worst case for frontier flood fill: the
    region's graph diameter is ~H*W/2 cells)."""
    g = np.full((H, W), 2, np.int8)
    i, j, d = 0, 0, 0
    dirs = [(0, 1), (1, 0), (0, -1), (-1, 0)]
    g[0, 0] = 1

    def free(a, b_):  # inside and not next to an older part of the corridor
        return 0 <= a < H and 0 <= b_ < W and g[a, b_] == 2

    turns = 0
    while turns < 2:
        di, dj = dirs[d]
        ni, nj = i + di, j + dj
        n2i, n2j = ni + di, nj + dj
        ok = free(ni, nj) and (not (0 <= n2i < H and 0 <= n2j < W) or g[n2i, n2j] == 2)
        if ok:  # also keep one cell of field between parallel arms
            li, lj = ni + dirs[(d + 3) % 4][0], nj + dirs[(d + 3) % 4][1]
            if 0 <= li < H and 0 <= lj < W and g[li, lj] == 1 and (li, lj) != (i, j):
                ok = False
        if ok:
            i, j = ni, nj
            g[i, j] = 1
            turns = 0
        else:
            d = (d + 1) % 4
            turns += 1
    return g


def floodfill_worst_case_compare(backend_cls, H, W):
    """FloodFill from both ends of a spiral corridor, from the field around it, and on a uniform grid (full-board
    region), point / 1x1-bbox / mask ingress; backend vs oracle."""
    g = spiral_grid(H, W)
    ones = np.argwhere(g == 1)
    seeds = [tuple(ones[0]), tuple(ones[-1]), tuple(np.argwhere(g == 2)[0]), (H - 1, W - 1), (H // 2, W // 2)]
    N = len(seeds) * 2
    ops = O.o2arc_ops()
    be, orc = backend_cls(N, H, W, -1, "o2arc", ops), OracleBackend(N, H, W, -1, "o2arc", ops)
    inp = np.stack([g] * len(seeds) + [np.full((H, W), 3, np.int8)] * len(seeds))
    dims = np.tile(np.array([[H, W]], np.int8), (N, 1))
    for b_ in (be, orc):
        b_.set_tasks(inp, dims, inp, dims)
        b_.reset()
    xy = np.array(seeds * 2, np.int32)
    errs = []
    for step, (ing, op) in enumerate((("point", 15), ("bbox", 17), ("mask", 10))):
        if ing == "point":
            pay = xy
        elif ing == "bbox":
            pay = np.concatenate([xy, xy], 1)
        else:
            pay = np.zeros((N, H, W), np.int8)
            pay[np.arange(N), xy[:, 0], xy[:, 1]] = 1
        opv = np.full(N, op, np.int32)
        be.step(ing, pay, opv)
        orc.step(ing, pay, opv)
        if not np.array_equal(be.get("grid"), orc.get("grid")):
            errs.append(f"{H}x{W} FloodFill step {step} ({ing}): grid differs for envs "
                        f"{np.nonzero((be.get('grid') != orc.get('grid')).reshape(N, -1).any(1))[0].tolist()}")
    assert int((orc.get("grid")[0] == 0).sum()) > H  # the corridor really was filled
    return errs
