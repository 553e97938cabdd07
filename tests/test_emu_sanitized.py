"""The kernel bodies compiled for the CPU — the one-wavefront-per-env step (arcle_amd/csrc/arcle_wave.h under tests/emu/wave_emu.cpp's lock-step
lane emulation) and the big-grid kernels (arcle_amd/csrc/arcle_big.h, tests/emu/big_emu.cpp) — under AddressSanitizer +
UndefinedBehaviorSanitizer: the workgroup's LDS is ONE heap block of exactly lds_bytes(PS, H), so a tile window that reads outside the
16 guard bytes / the neighbouring tiles / the block behind the last tile (shifted16's single clamp), a shift by the operand's width in the
packed-byte masks, or a count-zeros of 0 in the mask ingest aborts the child.  Results are compared with the oracle as everywhere else.
CPU only (sanitizers on the native test infrastructure)."""
import os
import subprocess
import sys

import pytest

import backends as B

CHILD = r"""
import sys
sys.path[:0] = [%(root)r, %(tests)r]
import backends as B
from oracle import oracle as O
from oracle import refdriver as RD
bad = []
for name in B.big_fixture_names():
    bad += B.replay_fixture(B.BigEmuBackend, name, max_steps=48)
for cls, H, W, kw in ((B.BigEmuTwoBackend, 40, 40, dict(flags=3)), (B.BigEmuTwoBackend, 33, 48, dict(flags=0)), (B.BigEmuOneBackend, 64, 33, dict(flags=3)),
                      (B.BigEmuBackend, 127, 127, dict(flags=1)), (B.BigEmuGenericBackend, 45, 45, dict(flags=3)), (B.BigEmuBackend, 127, 9, dict(flags=3)),
                      (B.BigEmuTwoBackend, 127, 16, dict(flags=3, int8_masks=True)), (B.BigEmuTwoBackend, 40, 48, dict(flags=3, new_forms=True)),
                      (B.BigEmuTwoBackend, 50, 50, dict(flags=3, int8_masks=True))):
    bad += B.random_trace_compare(cls, "o2arc", O.o2arc_ops(), H, W, N=3, S=40, seed=H * 5 + W, max_trial=3, bad_ops=True, **kw)
kind, ops = RD.variant_table("o2arc_exotic")
w = [1] * 35
for k in range(20, 28):
    w[k] = 4
bad += B.random_trace_compare(B.BigEmuTwoBackend, kind, ops, 45, 45, N=3, S=80, seed=6, op_weights=w, bad_ops=True)
bad += B.floodfill_worst_case_compare(B.BigEmuTwoBackend, 40, 40)
print("SANITIZED_OK" if not bad else "MISMATCH " + repr(bad[:5]))
"""


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_big_grid_kernel_bodies_under_asan_and_ubsan(tmp_path):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    lib = str(tmp_path / "libbig_emu_san.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-o", lib, os.path.join(B.ROOT, "tests", "emu", "big_emu.cpp")])
    preload = ":".join(x for x in (asan, _runtime("libubsan.so")) if x)
    env = dict(os.environ, LD_PRELOAD=preload, ARCLE_BIG_EMU_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = CHILD % {"root": B.ROOT, "tests": os.path.join(B.ROOT, "tests")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, f"rc {p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-6000:]}"
    assert "SANITIZED_OK" in p.stdout, p.stdout[-2000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]


WAVE_CHILD = r"""
import sys
sys.path[:0] = [%(root)r, %(tests)r]
import backends as B
from oracle import oracle as O
from oracle import refdriver as RD
bad = []
for name in B.fixture_names():
    bad += B.replay_fixture(B.EmuBackend, name, max_steps=40)
for kind, ops, H, W, kw in (("o2arc", O.o2arc_ops(), 30, 30, dict(flags=3)), ("o2arc", O.o2arc_ops(), 30, 30, dict(flags=3, new_forms=True)),
                            ("o2arc", O.o2arc_ops(), 17, 21, dict(flags=1)), ("o2arc", O.o2arc_ops(), 9, 13, dict(flags=3)), ("o2arc", O.o2arc_ops(), 2, 100, dict(flags=0)),
                            ("arc", O.arc_ops(), 30, 30, dict(flags=1)), ("raw", O.raw_ops(), 5, 5, dict(flags=0))):
    bad += B.random_trace_compare(B.EmuBackend, kind, ops, H, W, N=6, S=60, seed=H * 3 + W, max_trial=3, bad_ops=True, **kw)
kind, ops = RD.variant_table("o2arc_exotic")
bad += B.random_trace_compare(B.EmuBackend, kind, ops, 20, 17, N=6, S=80, seed=2, bad_ops=True)
bad += B.floodfill_worst_case_compare(B.EmuBackend, 30, 30)
print("SANITIZED_OK" if not bad else "MISMATCH " + repr(bad[:5]))
"""


def test_wave_kernel_body_under_asan_and_ubsan(tmp_path):
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    lib = str(tmp_path / "libwave_emu_san.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-o", lib, os.path.join(B.ROOT, "tests", "emu", "wave_emu.cpp")])
    preload = ":".join(x for x in (asan, _runtime("libubsan.so")) if x)
    env = dict(os.environ, LD_PRELOAD=preload, ARCLE_WAVE_EMU_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = WAVE_CHILD % {"root": B.ROOT, "tests": os.path.join(B.ROOT, "tests")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, f"rc {p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-6000:]}"
    assert "SANITIZED_OK" in p.stdout, p.stdout[-2000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
