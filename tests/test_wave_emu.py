"""The kernel body (arcle_amd/csrc/arcle_wave.h — the same header hipcc compiles for gfx950) executed by the
lock-step wavefront emulator of tests/emu, checked against the golden vectors and against the oracle.
This validates the kernel LOGIC on the CPU (plus: cross-lane ops only in wave-uniform control flow, values
declared uniform really are, stale LDS never matters).  The GPU parity tests proper are in test_hip_parity.py."""
import pytest

import backends as B
from oracle import oracle as O

OBJ_HEAVY = [1] * 10 + [2] * 10 + [4] * 8 + [2] * 3 + [1] * 4


@pytest.mark.parametrize("name", B.fixture_names())
def test_emulated_kernel_matches_golden(name):
    errs = B.replay_fixture(B.EmuBackend, name, max_steps=96)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("name", ["o2arc_30", "o2arc_10", "quirks_30", "o2arc_crop_10"])
def test_emulated_kernel_matches_golden_with_selected_elision(name):
    """ARCLE_STEP_ELIDE_SELECTED (skip the redundant zero-fill of `selected` for inactive envs) changes no state."""
    errs = B.replay_fixture(B.EmuBackend, name, max_steps=96, flags=B.STEP_ELIDE_SELECTED)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(30, 30), (10, 10), (5, 5), (3, 3), (7, 12), (12, 7), (32, 32), (1, 17), (20, 1), (16, 16)])
def test_emulated_kernel_vs_oracle_o2arc(H, W):
    for flags in (0, O.STEP_AUTORESET, B.STEP_ELIDE_SELECTED, O.STEP_AUTORESET | B.STEP_ELIDE_SELECTED):
        errs = B.random_trace_compare(B.EmuBackend, "o2arc", O.o2arc_ops(), H, W, N=6, S=40, seed=H * 100 + W + flags,
                                      max_trial=3 if flags else -1, flags=flags, op_weights=OBJ_HEAVY, bad_ops=True)
        assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("kind,ops", [("arc", O.arc_ops()), ("raw", O.raw_ops())])
def test_emulated_kernel_vs_oracle_other_kinds(kind, ops):
    errs = B.random_trace_compare(B.EmuBackend, kind, ops, 30, 30, N=6, S=40, seed=7, max_trial=3)
    assert not errs, "\n".join(errs[:10])


def exotic_table():
    """Rot180, Flip D0/D1, keep_sel, un-wrapped Color, wrapped Move, paste_blank=False, Crop — generators no shipped
    env installs (oracle/refdriver.py::variant_table, validated against the reference by diff_vs_reference.py)."""
    from oracle import refdriver as RD
    return RD.variant_table("o2arc_exotic")[1]


@pytest.mark.parametrize("H,W", [(30, 30), (16, 16), (20, 17), (17, 20), (12, 12), (9, 32)])
def test_emulated_kernel_vs_oracle_exotic_ops(H, W):
    w = [1] * 20 + [4] * 8 + [2] * 7
    errs = B.random_trace_compare(B.EmuBackend, "o2arc", exotic_table(), H, W, N=6, S=48, seed=3 * H + W, op_weights=w)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(30, 30), (10, 10), (5, 7)])
def test_emulated_reset_from_task_table(H, W):
    errs = B.task_table_compare(B.EmuBackend, H, W, N=12, T=9, seed=H)
    assert not errs, "\n".join(errs)


@pytest.mark.parametrize("H,W,ingress", [(30, 30, "bbox"), (10, 10, "bbox"), (30, 30, "point"), (5, 7, "bbox")])
def test_emulated_rollout_equals_sequential_steps(H, W, ingress):
    for flags in (0, O.STEP_AUTORESET, B.STEP_ELIDE_SELECTED):
        errs = B.rollout_compare(B.EmuBackend, "o2arc", O.o2arc_ops(), H, W, N=6, T=40, seed=H + W + flags,
                                 ingress=ingress, flags=flags)
        assert not errs, "\n".join(errs[:10])


def test_emulated_rollout_other_kinds():
    for kind, ops in (("arc", O.arc_ops()), ("raw", O.raw_ops())):
        errs = B.rollout_compare(B.EmuBackend, kind, ops, 30, 30, N=4, T=30, seed=9)
        assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(30, 30), (32, 32), (17, 21), (12, 12), (6, 40)])
def test_emulated_floodfill_worst_case(H, W):
    errs = B.floodfill_worst_case_compare(B.EmuBackend, H, W)
    assert not errs, "\n".join(errs)
