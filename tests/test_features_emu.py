"""CPU suite: the round-2 features of the kernels (run lock-step by the wave emulator, tests/emu) against the golden
vectors captured from the reference (tests/golden/make_golden_research.py) and the host mirror of the device sampler."""
import pytest

import backends as B
import features as F
import rows as R


@pytest.mark.parametrize("check", [F.wrappers, F.augment, F.dense, F.reset_on_submit, F.flat, F.continue_rule, F.sampler, F.resample_content,
                                   F.truncation, F.packed, F.bad_selection], ids=lambda f: f.__name__)
def test_emulated_kernel_feature(check):
    errs = check(B.EmuBackend)
    assert not errs, "\n".join(errs[:10])


@pytest.mark.parametrize("check", [R.new_ingress_forms, R.mask_bits_packer, R.state_rows_roundtrip, R.transition_rows, R.flat_tail,
                                   R.dense_on_autoreset, R.incremental_rows, R.dense_cache, R.partial_store_invariants], ids=lambda f: f.__name__)
def test_emulated_round3_boundary(check):
    """bbox5 / bit-packed ingress, state rows in, stateless batched transition, row tail, dense on auto-reset steps — the kernel
    bodies run lock-step on the CPU against the oracle."""
    errs = check(B.EmuBackend)
    assert not errs, "\n".join(errs[:10])
