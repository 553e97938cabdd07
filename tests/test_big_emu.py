"""Grids beyond 1024 cells (H * W > ARCLE_MAX_CELLS; the reference takes any max_grid_size, base.py:37-49) — CPU side:
  * the oracle against golden vectors captured from the UNMODIFIED reference at those sizes (tests/golden/make_golden_big.py);
  * the workgroup-per-env kernel bodies (arcle_amd/csrc/arcle_big.h, run on host threads by tests/emu/big_emu.cpp) against the oracle and
    against the same golden vectors.
The GPU side is tests/test_big_hip.py."""
import numpy as np
import pytest

import backends as B
from oracle import oracle as O
from oracle import refdriver as RD


@pytest.mark.parametrize("name", B.big_fixture_names())
def test_oracle_reproduces_big_golden(name):
    errs = B.replay_fixture(B.OracleBackend, name)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("name", B.big_fixture_names())
def test_big_emulator_reproduces_big_golden(name):
    errs = B.replay_fixture(B.BigEmuBackend, name, max_steps=96)
    assert not errs, "\n".join(errs[:10])


def test_big_fixture_set_is_complete():
    assert len(B.big_fixture_names()) == 8


@pytest.mark.parametrize("H,W", [(40, 40), (33, 48), (64, 64), (127, 9), (100, 20)])
@pytest.mark.parametrize("kind,flags,max_trial", [("o2arc", 0, -1), ("o2arc", 3, 3), ("arc", 1, 3), ("raw", 0, 2)])
def test_big_emulator_random_traces_vs_oracle(H, W, kind, flags, max_trial):
    errs = B.random_trace_compare(B.BigEmuBackend, kind, O.KIND_OPS[kind](), H, W, N=5, S=40, seed=H * 131 + W + flags, max_trial=max_trial,
                                  flags=flags, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W,ingress_mix", [(33, 32, True), (40, 40, False)])
def test_big_emulator_one_chunk_per_thread_instantiation(H, W, ingress_mix):
    """a LEAN kernel of the product (ARCLE_BIG_CPT=1): every "my chunks" loop is a single guarded body"""
    errs = B.random_trace_compare(B.BigEmuOneBackend, "o2arc", O.o2arc_ops(), H, W, N=3, S=40, seed=H + W, flags=3, max_trial=3, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W,kind,flags", [(40, 40, "o2arc", 3), (33, 48, "o2arc", 0), (64, 64, "o2arc", 3), (100, 20, "arc", 1), (127, 16, "o2arc", 3)])
def test_big_emulator_two_chunks_per_thread_instantiation(H, W, kind, flags):
    """the LEAN kernel the product's launcher picks by default: two guarded bodies per "my chunks" loop, a workgroup of half as many threads as
    the plane has chunks (the last thread's second chunk may lie beyond the plane)"""
    errs = B.random_trace_compare(B.BigEmuTwoBackend, kind, O.KIND_OPS[kind](), H, W, N=3, S=40, seed=2 * H + W, flags=flags, max_trial=3, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


def test_big_emulator_two_chunks_per_thread_mask_forms_and_fill():
    """... under the mask ingress forms (int8 and bit-packed rows) and through the flood fill's worst case"""
    errs = B.random_trace_compare(B.BigEmuTwoBackend, "o2arc", O.o2arc_ops(), 40, 48, N=3, S=30, seed=5, max_trial=3, flags=3, new_forms=True)
    assert not errs, "\n".join(errs[:10])
    errs = B.floodfill_worst_case_compare(B.BigEmuTwoBackend, 40, 40)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("backend", ["BigEmuBackend", "BigEmuTwoBackend", "BigEmuGenericBackend"])
@pytest.mark.parametrize("H,W", [(40, 40), (33, 48), (100, 20), (127, 9)])
def test_big_emulator_int8_masks_of_any_value(backend, H, W):
    """masks of arbitrary int8 values: the whole-word reductions of the mask ingest (non-zero map, rows / columns by count-zeros, dot-product
    sum, "first 1" arg-max with the per-cell chain behind it for chunks that hold other values) against the oracle's per-cell loops"""
    w = [3] * 10 + [3] * 10 + [2] * 8 + [2] * 7  # (Color / FloodFill weighted up: they read sum and arg-max)
    errs = B.random_trace_compare(getattr(B, backend), "o2arc", O.o2arc_ops(), H, W, N=4, S=40, seed=H + 3 * W, flags=3, max_trial=3, op_weights=w, int8_masks=True)
    assert not errs, "\n".join(errs[:10])


def test_big_emulator_four_chunks_per_thread_instantiation():
    errs = B.random_trace_compare(B.BigEmuFourBackend, "o2arc", O.o2arc_ops(), 64, 48, N=3, S=40, seed=31, flags=3, max_trial=3, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


def test_big_emulator_generic_instantiation_still_agrees():
    errs = B.random_trace_compare(B.BigEmuGenericBackend, "o2arc", O.o2arc_ops(), 40, 40, N=4, S=40, seed=77, flags=3, max_trial=3, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


def test_big_emulator_127x127():
    errs = B.random_trace_compare(B.BigEmuBackend, "o2arc", O.o2arc_ops(), 127, 127, N=3, S=30, seed=9)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("variant", ["o2arc_exotic", "o2arc_crop"])
def test_big_emulator_exotic_tables(variant):
    """Rotate 180, Flip D0 / D1, keep_sel- / reset_sel-wrapped ops, Paste without blanks, CropGrid — object ops weighted up."""
    kind, ops = RD.variant_table(variant)
    w = [1] * 35
    for k in range(20, 28):
        w[k] = 4
    errs = B.random_trace_compare(B.BigEmuBackend, kind, ops, 45, 45, N=5, S=100, seed=len(variant), op_weights=w, bad_ops=True)
    assert not errs, "\n".join(errs[:10])


def test_big_emulator_floodfill_worst_case():
    """a one-cell-wide spiral corridor: the row-board fill needs one pass per row of every vertical leg"""
    errs = B.floodfill_worst_case_compare(B.BigEmuBackend, 40, 40)
    assert not errs, "\n".join(errs[:10])


import bigcases as C  # noqa: E402


@pytest.mark.parametrize("kind", ["o2arc", "arc", "raw"])
def test_big_emulator_rows(kind):
    C.rows_case(B.BigEmuBackend, kind)


def test_big_emulator_truncation_and_autoreset():
    C.truncation_case(B.BigEmuBackend)


def test_big_emulator_task_table_and_device_draw():
    C.task_table_case(B.BigEmuBackend)


def test_big_emulator_transition_rows():
    """the stateless batched transition (o2arcenv.py:149-151) on grids of more than 1024 cells: rows in, scratch envs, rows out"""
    import rows as R
    errs = R.transition_rows(B.BigEmuBackend, cases=(("o2arc", 40, 40, 3), ("arc", 36, 41, 3), ("raw", 35, 30, 2)))
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("H,W", [(40, 40), (50, 50)])  # (square planes: a Rotate out of its domain is a skipped step too, and rows.dense_on_autoreset does not model those)
def test_big_emulator_dense_pair(H, W):
    """ARCLE_STEP_DENSE (agents/env.py:44-58 as an exact integer pair) incl. the (0, 0) of auto-reset and skipped steps"""
    import rows as R
    errs = R.dense_on_autoreset(B.BigEmuBackend, H, W)
    assert not errs, "\n".join(errs[:10])


def test_big_emulator_task_augmentation():
    C.aug_case(B.BigEmuBackend)


@pytest.mark.parametrize("H,W", [(40, 40), (36, 41)])
def test_big_emulator_record_and_bit_packed_forms(H, W):
    """5-tuple records and bit-packed boolean masks (rows of plane_stride / 8 bytes) against the oracle's classic forms; the packer"""
    errs = B.random_trace_compare(B.BigEmuBackend, "o2arc", O.o2arc_ops(), H, W, N=5, S=60, seed=H + W, max_trial=3, flags=3, new_forms=True)
    assert not errs, "\n".join(errs[:10])
    rng = np.random.default_rng(1)
    be = B.BigEmuBackend(5, H, W, 3, "o2arc", O.o2arc_ops())
    m = (rng.random((5, H, W)) < 0.3).astype(np.int8) * rng.integers(-3, 4, (5, H, W)).astype(np.int8)
    assert np.array_equal(be.pack_mask_bits(m), B.pack_bits(m))


def test_big_emulator_byte_accounting():
    """arcle_enable_accounting on the big path: every 16-byte access a thread issues is counted.  CopyFromInput on an inactive env with the
    zero-fill of `selected` elided reads the input plane and writes the grid plane: 2 planes + the env's scalars."""
    H, W, N = 40, 40, 3
    be = B.BigEmuBackend(N, H, W, 3, "o2arc", O.o2arc_ops())
    inp = np.random.default_rng(0).integers(0, 10, (N, H, W)).astype(np.int8)
    dims = np.full((N, 2), H, np.int8)
    be.set_tasks(inp, dims, inp, dims)
    be.reset()
    be.count_bytes = True
    be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 31, np.int32), 2)
    scal = 2 * 16 + 16 + 20 + 5
    assert be.acct[N:].tolist() == [2 * be.PS + scal] * N and be.acct[:N].tolist() == [2 * H * W + scal] * N
    be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 3, np.int32), 2)  # Color on a one-cell selection: 1 grid chunk read + written
    assert be.acct[N:].tolist() == [2 * be.PS + 32 + 2 * scal] * N
