"""CPU suite: the round-2 features of the kernels (run lock-step by the wave emulator, tests/emu) against the golden
vectors captured from the reference (tests/golden/make_golden_research.py) and the host mirror of the device sampler."""
import pytest

import backends as B
import features as F


@pytest.mark.parametrize("check", [F.wrappers, F.augment, F.dense, F.reset_on_submit, F.flat, F.continue_rule, F.sampler,
                                   F.truncation, F.packed, F.bad_selection], ids=lambda f: f.__name__)
def test_emulated_kernel_feature(check):
    errs = check(B.EmuBackend)
    assert not errs, "\n".join(errs[:10])
