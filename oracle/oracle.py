"""ctypes front-end of oracle/arcle_oracle.c — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see the header of arcle_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by arcle_amd/.

The state arrays are NumPy arrays in exactly the layout include/arcle_hip.h declares for the
device (planes int8 [N,H,W], rec int8 [N,16], cnt int32 [N,2]), so a device state copied to
the host can be compared array-for-array.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARCLE_ORACLE_LIB: load another build of the same source instead (the ASan / UBSan build of tests/test_oracle_sanitized.py)
_LIB_PATH = os.environ.get("ARCLE_ORACLE_LIB") or os.path.join(_HERE, "libarcle_oracle.so")

PLANES = ["input", "grid", "selected", "clip", "object", "object_sel", "background", "answer"]
REC = {  # name -> (byte offset, length) in rec[env]
    "input_dim": (0, 2), "grid_dim": (2, 2), "clip_dim": (4, 2), "object_dim": (6, 2),
    "object_pos": (8, 2), "trials_remain": (10, 1), "terminated": (11, 1), "active": (12, 1),
    "rotation_parity": (13, 1), "answer_dim": (14, 2),
}
STEP_AUTORESET = 1
ST_BAD_OP, ST_ROTATE_DOMAIN = 1, 2

# op kinds (include/arcle_hip.h enum arcle_op_kind)
(OP_NONE, OP_COLOR, OP_FLOODFILL, OP_MOVE, OP_ROTATE, OP_FLIP, OP_COPY, OP_PASTE, OP_COPY_FROM_INPUT,
 OP_RESET_GRID, OP_RESIZE_GRID, OP_CROP_GRID, OP_RESIZE_TO_ANSWER, OP_SUBMIT) = range(14)
F_RESET_SEL, F_KEEP_SEL = 1, 2


def desc(kind, arg=0, flags=0):
    return kind | (arg << 8) | (flags << 16)


def o2arc_ops():
    """Op table of O2ARCv2Env.create_operations (/root/reference/arcle/envs/o2arcenv.py:88-113)."""
    R = F_RESET_SEL
    ops = [desc(OP_COLOR, c, R) for c in range(10)]            # :91
    ops += [desc(OP_FLOODFILL, c, R) for c in range(10)]       # :92
    ops += [desc(OP_MOVE, d) for d in range(4)]                # :95
    ops += [desc(OP_ROTATE, 1), desc(OP_ROTATE, 3)]            # :96-97
    ops += [desc(OP_FLIP, 0), desc(OP_FLIP, 1)]                # :98-99
    ops += [desc(OP_COPY, 0, R), desc(OP_COPY, 1, R), desc(OP_PASTE, 1, R)]  # :102-104
    ops += [desc(OP_COPY_FROM_INPUT, 0, R), desc(OP_RESET_GRID, 0, R), desc(OP_RESIZE_GRID, 0, R)]  # :107-109
    ops += [desc(OP_SUBMIT)]                                   # :112
    return ops


def arc_ops():
    """The 27 ops ARCEnv.create_operations fills in (/root/reference/arcle/envs/arcenv.py:123-137);
    the reference class itself cannot be constructed (8 None slots)."""
    ops = [desc(OP_COLOR, c) for c in range(10)]
    ops += [desc(OP_FLOODFILL, c) for c in range(10)]
    ops += [desc(OP_COPY, 0), desc(OP_COPY, 1), desc(OP_PASTE, 1)]
    ops += [desc(OP_COPY_FROM_INPUT), desc(OP_RESET_GRID), desc(OP_RESIZE_GRID), desc(OP_SUBMIT)]
    return ops


def raw_ops():
    """RawARCEnv.create_operations (/root/reference/arcle/envs/arcenv.py:26-41)."""
    return [desc(OP_COLOR, c) for c in range(10)] + [desc(OP_RESIZE_TO_ANSWER), desc(OP_SUBMIT)]


KIND_PLANES = {
    "o2arc": PLANES,
    "arc": ["input", "grid", "clip", "answer"],
    "raw": ["input", "grid", "answer"],
}
KIND_OPS = {"o2arc": o2arc_ops, "arc": arc_ops, "raw": raw_ops}


def build(force=False):
    """Compiles arcle_oracle.c into libarcle_oracle.so next to it (gcc only)."""
    src = os.path.join(_HERE, "arcle_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "arcle_hip.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-Wall", "-o", _LIB_PATH, src])
    return _LIB_PATH


def build_sanitized(force=False):
    """The same source under AddressSanitizer + UndefinedBehaviorSanitizer (gcc), as libarcle_oracle_san.so next to it: the C
    restatement is pointer arithmetic over int8 planes, the sanitizers check what the golden vectors cannot (reads past a plane,
    signed overflow, misaligned access).  Loaded through ARCLE_ORACLE_LIB with libasan preloaded (tests/test_oracle_sanitized.py)."""
    src = os.path.join(_HERE, "arcle_oracle.c")
    out = os.path.join(_HERE, "libarcle_oracle_san.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fopenmp",
                           "-fPIC", "-shared", "-Wall", "-o", out, src])
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("ARCLE_ORACLE_LIB"):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_create.restype = ctypes.c_void_p
        L.oracle_create.argtypes = [ctypes.c_int] * 4
        L.oracle_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_bind.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_set_op_table.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_get_status.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.oracle_get_status.restype = ctypes.c_uint32
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        for name in ("oracle_step_mask", "oracle_step_bbox", "oracle_step_point"):
            getattr(L, name).argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_uint32]
        _lib = L
    return _lib


def set_threads(n):
    """Host threads used by the env loops of the step entry points (OpenMP); 1 = serial (default)."""
    lib().oracle_set_threads(int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleEnv:
    """N independent envs stepped by the C restatement.  `kind` in {'o2arc','arc','raw'} selects
    which planes exist and the default op table; `ops` overrides the table."""

    def __init__(self, n_envs, H, W, max_trial=-1, kind="o2arc", ops=None):
        self.N, self.H, self.W, self.max_trial, self.kind = n_envs, H, W, max_trial, kind
        self.planes = {k: np.zeros((n_envs, H, W), np.int8) for k in KIND_PLANES[kind]}
        self.rec = np.zeros((n_envs, 16), np.int8)
        self.cnt = np.zeros((n_envs, 2), np.int32)
        self._h = lib().oracle_create(n_envs, H, W, max_trial)
        if not self._h:
            raise ValueError("unsupported configuration")
        arr = (ctypes.c_void_p * 8)(*[_p(self.planes.get(k)) for k in PLANES])
        self._keep = arr
        assert lib().oracle_bind(self._h, arr, _p(self.rec), _p(self.cnt)) == 0
        self.ops = list(ops if ops is not None else KIND_OPS[kind]())
        t = np.asarray(self.ops, np.uint32)
        rc = lib().oracle_set_op_table(self._h, _p(t), len(t))
        if rc != 0:
            raise ValueError(f"op table rejected ({rc})")
        self.reward = np.zeros(n_envs, np.int32)
        self.term = np.zeros(n_envs, np.uint8)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_destroy(self._h)
            self._h = None

    # -- record accessors -------------------------------------------------------------------
    def field(self, name):
        off, n = REC[name]
        return self.rec[:, off:off + n]

    def set_tasks(self, inputs, answers):
        """inputs / answers: lists (len N) of 2-D int8 arrays (un-padded), as Loader.pick returns."""
        for n, (a, b) in enumerate(zip(inputs, answers)):
            self.planes["input"][n] = 0
            self.planes["input"][n, :a.shape[0], :a.shape[1]] = a
            self.planes["answer"][n] = 0
            self.planes["answer"][n, :b.shape[0], :b.shape[1]] = b
            self.field("input_dim")[n] = a.shape
            self.field("answer_dim")[n] = b.shape

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().oracle_reset(self._h, _p(m))

    def status(self, clear=True):
        return lib().oracle_get_status(self._h, int(clear))

    def _step(self, fn, sel, op, flags):
        op = np.ascontiguousarray(op, np.int32)
        fn(self._h, _p(sel), _p(op), _p(self.reward), _p(self.term), flags)
        return self.reward, self.term

    def step_mask(self, sel, op, flags=0):
        sel = np.ascontiguousarray(np.asarray(sel).astype(np.int8, copy=False)).reshape(self.N, self.H, self.W)
        return self._step(lib().oracle_step_mask, sel, op, flags)

    def step_bbox(self, bbox, op, flags=0):
        return self._step(lib().oracle_step_bbox, np.ascontiguousarray(bbox, np.int32).reshape(self.N, 4), op, flags)

    def step_point(self, xy, op, flags=0):
        return self._step(lib().oracle_step_point, np.ascontiguousarray(xy, np.int32).reshape(self.N, 2), op, flags)

    def state_dict(self, n):
        """State of env n in the reference's dict structure (o2arcenv.py:16-34)."""
        f = lambda k: self.field(k)[n].copy()
        d = {"trials_remain": f("trials_remain"), "terminated": f("terminated"),
             "input": self.planes["input"][n].copy(), "input_dim": f("input_dim"),
             "grid": self.planes["grid"][n].copy(), "grid_dim": f("grid_dim")}
        if "clip" in self.planes:
            d["clip"] = self.planes["clip"][n].copy()
            d["clip_dim"] = f("clip_dim")
        if "selected" in self.planes:
            d["selected"] = self.planes["selected"][n].copy()
            d["object_states"] = {
                "active": f("active"), "object": self.planes["object"][n].copy(),
                "object_sel": self.planes["object_sel"][n].copy(), "object_dim": f("object_dim"),
                "object_pos": f("object_pos"), "background": self.planes["background"][n].copy(),
                "rotation_parity": f("rotation_parity")}
        return d
