"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on the native test
infrastructure).  A child interpreter loads the instrumented build (ARCLE_ORACLE_LIB, libasan preloaded) and replays every golden
fixture plus random traces with out-of-contract inputs (int8 wrap-around, off-grid objects, bad op indices, all ingress forms); any
finding aborts the child (-fno-sanitize-recover).  CPU only."""
import os
import subprocess
import sys

import pytest

import backends as B
from oracle import oracle as O

CHILD = r"""
import sys
sys.path[:0] = [%(root)r, %(tests)r]
import numpy as np
import backends as B
from oracle import oracle as O
assert O._LIB_PATH.endswith("libarcle_oracle_san.so"), O._LIB_PATH
bad = []
for name in B.fixture_names():
    bad += B.replay_fixture(B.OracleBackend, name)
# random traces: the oracle against itself is vacuous as a comparison — the point is that the sanitizers watch every access
from oracle import refdriver as RD
for kind, ops, H, W in (("o2arc", O.o2arc_ops(), 30, 30), ("o2arc", O.o2arc_ops(), 7, 12), ("o2arc", O.o2arc_ops(), 2, 100),
                        ("o2arc", RD.variant_table("o2arc_exotic")[1], 17, 20), ("arc", O.arc_ops(), 30, 30), ("raw", O.raw_ops(), 5, 5)):
    bad += B.random_trace_compare(B.OracleBackend, kind, ops, H, W, N=24, S=120, seed=H * 7 + W, max_trial=3, flags=O.STEP_AUTORESET, bad_ops=True)
for H, W in ((30, 30), (17, 21), (6, 40)):
    bad += B.floodfill_worst_case_compare(B.OracleBackend, H, W)
O.set_threads(4)
bad += B.random_trace_compare(B.OracleBackend, "o2arc", O.o2arc_ops(), 30, 30, N=64, S=40, seed=3, max_trial=-1)
print("SANITIZED_OK" if not bad else "MISMATCH " + repr(bad[:5]))
"""


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    lib = O.build_sanitized()
    preload = ":".join(x for x in (asan, _runtime("libubsan.so")) if x)
    env = dict(os.environ, LD_PRELOAD=preload, ARCLE_ORACLE_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=77",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="4")
    code = CHILD % {"root": B.ROOT, "tests": os.path.join(B.ROOT, "tests")}
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, f"rc {p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-4000:]}"
    assert "SANITIZED_OK" in p.stdout, p.stdout[-2000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
