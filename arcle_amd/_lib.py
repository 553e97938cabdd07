"""ctypes binding of libarcle_hip.so — the C ABI declared in include/arcle_hip.h.

The HIP library is THE product path: there is no CPU fallback.  If the shared object is missing
or the machine has no HIP device, creating an env raises `ArcleHipError` loudly.
"""
import ctypes
import os
import shutil
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("ARCLE_HIP_LIB") or os.path.join(_CSRC, "libarcle_hip.so")  # override: A/B kernel tuning
# two translation units: the one-wavefront-per-env kernels + the C ABI (arcle_hip.hip <- arcle_wave.h) and the workgroup-per-env kernels
# for grids of more than 1024 cells (arcle_big.hip <- arcle_big.h); arcle_big_params.h is shared
UNITS = [os.path.join(_CSRC, "arcle_hip.hip"), os.path.join(_CSRC, "arcle_big.hip")]
SOURCES = UNITS + [os.path.join(_CSRC, "arcle_wave.h"), os.path.join(_CSRC, "arcle_big.h"), os.path.join(_CSRC, "arcle_big_params.h"),
                   os.path.join(_CSRC, "..", "..", "include", "arcle_hip.h")]

ABI_VERSION = 5
N_PLANES = 8
MAX_OPS = 64
BITS_STRIDE = 128  # bytes between envs of a bit-packed mask array (ARCLE_MAX_CELLS / 8)
INGRESS = {"mask": 0, "bbox": 1, "point": 2, "bbox5": 3, "bits": 4}  # enum arcle_ingress
EXPORTS = ["arcle_abi_version", "arcle_create", "arcle_destroy", "arcle_get_buffers", "arcle_set_op_table",
           "arcle_can_elide_selected", "arcle_reset", "arcle_set_task_table", "arcle_reset_from_table",
           "arcle_step_mask", "arcle_step_bbox", "arcle_step_point", "arcle_step_bbox5", "arcle_step_bits", "arcle_pack_mask_bits", "arcle_mask_bits_stride",
           "arcle_step_many", "arcle_set_dispatch_order", "arcle_hint_next_ops", "arcle_launch_info", "arcle_autotune", "arcle_rollout_bbox", "arcle_rollout_point", "arcle_rollout_mask", "arcle_set_truncation",
           "arcle_packed_obs_size", "arcle_pack_obs", "arcle_set_packed_output", "arcle_set_sampler", "arcle_reset_sampled",
           "arcle_reset_from_table_aug", "arcle_set_dense_output", "arcle_invalidate", "arcle_flat_obs_size", "arcle_flatten_obs",
           "arcle_set_flat_output", "arcle_set_flat_output_ex", "arcle_set_flat_seq", "arcle_get_state_rows", "arcle_set_state_rows",
           "arcle_transition_rows", "arcle_get_plane", "arcle_set_plane", "arcle_get_status",
           "arcle_enable_accounting", "arcle_get_accounting", "arcle_get_accounting_ex", "arcle_last_error"]


class ArcleHipError(RuntimeError):
    pass


class Config(ctypes.Structure):
    _fields_ = [("n_envs", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("max_trial", ctypes.c_int32), ("device", ctypes.c_int32), ("plane_stride", ctypes.c_int32)]


class Buffers(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_void_p * N_PLANES), ("rec", ctypes.c_void_p), ("cnt", ctypes.c_void_p)]


def build(force=False, verbose=False):
    """Compiles csrc/arcle_hip.hip for gfx950 into csrc/libarcle_hip.so (in-tree)."""
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in SOURCES):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise ArcleHipError("hipcc not found: cannot build libarcle_hip.so")
    # -amdgpu-kernarg-preload-count: the step kernel's leading scalar arguments (the four per-env array bases, batch size, launch
    # shape) are in SGPRs when a wave starts instead of behind a scalar load of the argument block
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]  # (run-time chunk loops carry an unroll request meant for the LEAN instantiations)
    objs = [os.path.join(_CSRC, os.path.basename(u)[:-4] + ".o") for u in UNITS]
    cmds = [common + ["-mllvm", "-amdgpu-kernarg-preload-count=13", "-c", UNITS[0], "-o", objs[0]],
            # -amdgpu-atomic-optimizer-strategy=DPP: the big-grid kernels reduce a selection (any / sum / arg-max / bounding box) with LDS
            # atomics on ONE address per value; the compiler's default rewrites each into a scalar loop over the active lanes (~8 scalar
            # instructions a lane: 1280 instead of 350 scalar instructions a wave, mask-ingress steps twice as slow), DPP into a
            # cross-lane reduction + one atomic per wave (profiles/round6_experiments.txt §2f)
            common + ["-mllvm", "-amdgpu-atomic-optimizer-strategy=DPP", "-c", UNITS[1], "-o", objs[1]]]
    if verbose:
        for c in cmds:
            print(" ".join(c))
    procs = [subprocess.Popen(c) for c in cmds]  # (the two units compile side by side: the first takes minutes, the second seconds)
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise ArcleHipError(f"hipcc failed ({rcs})")
    link = common + ["-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    for o in objs:
        os.remove(o)
    return LIB_PATH


_lib = None


def lib():
    """Loads the shared library (does not need a GPU; creating a handle does)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArcleHipError(
            f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  arcle_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32
    L.arcle_abi_version.restype = ctypes.c_int
    L.arcle_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(Buffers), ctypes.POINTER(vp)]
    L.arcle_destroy.argtypes = [vp]
    L.arcle_get_buffers.argtypes = [vp, ctypes.POINTER(Buffers)]
    L.arcle_set_op_table.argtypes = [vp, ctypes.POINTER(u32), i32]
    L.arcle_can_elide_selected.argtypes = [vp]
    L.arcle_reset.argtypes = [vp, vp, vp]
    L.arcle_set_task_table.argtypes = [vp, vp, vp, vp, vp, i32]
    L.arcle_reset_from_table.argtypes = [vp, vp, vp, vp]
    for name in ("arcle_step_mask", "arcle_step_bbox", "arcle_step_point"):
        getattr(L, name).argtypes = [vp, vp, vp, vp, vp, u32, vp]
    L.arcle_step_bbox5.argtypes = [vp, vp, vp, vp, u32, vp]
    L.arcle_step_bits.argtypes = [vp, vp, vp, vp, vp, u32, vp]
    L.arcle_pack_mask_bits.argtypes = [vp, vp, vp, vp]
    L.arcle_mask_bits_stride.argtypes = [vp]
    L.arcle_step_many.argtypes = [vp, ctypes.c_int, i32, vp, vp, vp, vp, u32, vp]
    L.arcle_set_dispatch_order.argtypes = [vp, ctypes.c_int]
    L.arcle_hint_next_ops.argtypes = [vp, vp, i32]
    L.arcle_launch_info.argtypes = [vp, ctypes.c_int, u32, vp]
    L.arcle_autotune.argtypes = [vp, ctypes.c_int, i32, vp, vp, u32, vp, i32, vp]
    L.arcle_set_flat_output_ex.argtypes = [vp, vp, i32, ctypes.c_int, ctypes.c_int]
    L.arcle_set_flat_seq.argtypes = [vp, i32]
    L.arcle_get_state_rows.argtypes = [vp, vp, i32, vp]
    L.arcle_set_state_rows.argtypes = [vp, vp, i32, vp, vp]
    L.arcle_transition_rows.argtypes = [vp, i32, vp, i32, ctypes.c_int, vp, vp, vp, vp, i32, ctypes.c_int, vp, vp, u32, vp]
    L.arcle_get_plane.argtypes = [vp, ctypes.c_int, vp, vp]
    L.arcle_set_plane.argtypes = [vp, ctypes.c_int, vp, vp]
    L.arcle_get_accounting_ex.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                          ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, vp]
    L.arcle_set_truncation.argtypes = [vp, vp, i32]
    L.arcle_set_sampler.argtypes = [vp, vp, vp, i32, ctypes.c_uint64, ctypes.c_int64, vp, vp, u32]
    L.arcle_reset_sampled.argtypes = [vp, vp, vp]
    L.arcle_reset_from_table_aug.argtypes = [vp, vp, vp, vp, vp, vp]
    L.arcle_set_dense_output.argtypes = [vp, vp]
    L.arcle_invalidate.argtypes = [vp, vp]
    L.arcle_packed_obs_size.argtypes = [vp]
    L.arcle_pack_obs.argtypes = [vp, vp, vp, vp, vp]
    L.arcle_set_packed_output.argtypes = [vp, vp]
    for name in ("arcle_rollout_bbox", "arcle_rollout_point", "arcle_rollout_mask"):
        getattr(L, name).argtypes = [vp, i32, vp, vp, vp, vp, u32, vp]
    L.arcle_flat_obs_size.argtypes = [vp, ctypes.c_int]
    L.arcle_flatten_obs.argtypes = [vp, vp, i32, ctypes.c_int, vp]
    L.arcle_set_flat_output.argtypes = [vp, vp, i32, ctypes.c_int]
    L.arcle_get_status.argtypes = [vp, ctypes.POINTER(u32), ctypes.c_int, vp]
    L.arcle_enable_accounting.argtypes = [vp, ctypes.c_int]
    L.arcle_get_accounting.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.c_int, vp]
    L.arcle_last_error.argtypes = [vp]
    L.arcle_last_error.restype = ctypes.c_char_p
    if L.arcle_abi_version() != ABI_VERSION:
        raise ArcleHipError("libarcle_hip.so ABI version mismatch — rebuild")
    _lib = L
    return L
