"""The C-ABI library: builds for gfx950, loads, exports every symbol include/arcle_hip.h declares, and
refuses loudly to work without a HIP device (no CPU fallback).  No compute calls — CPU only."""
import ctypes
import os
import re

import pytest

import backends as B
from arcle_amd import _lib, actions
from oracle import oracle as O

HEADER = os.path.join(B.ROOT, "include", "arcle_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(arcle_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    assert set(declared_functions()) == set(_lib.EXPORTS)


def test_library_builds_and_exports_every_declared_symbol():
    path = _lib.build()
    assert os.path.exists(path)
    L = ctypes.CDLL(path)
    for name in declared_functions():
        assert hasattr(L, name), f"libarcle_hip.so does not export {name}"
    assert L.arcle_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from arcle_amd.engine import EnvBatch
    with pytest.raises(_lib.ArcleHipError):
        EnvBatch(4, 5, 5)
    # the C entry point itself reports the missing device instead of computing anything
    L = _lib.lib()
    cfg = _lib.Config(4, 5, 5, -1, -1, 0)
    h = ctypes.c_void_p()
    assert L.arcle_create(ctypes.byref(cfg), None, ctypes.byref(h)) == -4  # ARCLE_ERR_NO_DEVICE
    assert not h.value


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(B.ROOT, "arcle_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libarcle_oracle" not in txt, f


def test_descriptor_encoding_agrees_with_oracle_tables():
    from arcle_amd.envs import O2ARCv2Env, ARCEnv, RawARCEnv
    assert actions.table_descs(O2ARCv2Env.default_operations()) == O.o2arc_ops()
    assert actions.table_descs(ARCEnv.default_operations()) == O.arc_ops()
    assert actions.table_descs(RawARCEnv.default_operations()) == O.raw_ops()
