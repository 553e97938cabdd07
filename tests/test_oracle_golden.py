"""The oracle (oracle/arcle_oracle.c) against the golden vectors captured from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import backends as B
from oracle import oracle as O


@pytest.mark.parametrize("name", B.fixture_names())
def test_oracle_matches_golden(name):
    errs = B.replay_fixture(B.OracleBackend, name)
    assert not errs, "\n".join(errs[:10])


def test_fixture_inventory():
    names = B.fixture_names()
    for required in ("o2arc_30", "o2arc_10", "o2arc_05", "arc_30", "raw_05", "quirks_30", "o2arc_crop_10",
                     "o2arc_exotic_12"):
        assert required in names


def test_op_tables_match_reference_numbering():
    # o2arcenv.py:88-113 — independent confirmation: tests/o2arc_check.py:21-99 of the reference
    ops = O.o2arc_ops()
    assert len(ops) == 35 and len(O.arc_ops()) == 27 and len(O.raw_ops()) == 12
    kinds = [d & 0xFF for d in ops]
    assert kinds[0:10] == [O.OP_COLOR] * 10 and kinds[10:20] == [O.OP_FLOODFILL] * 10
    assert kinds[20:24] == [O.OP_MOVE] * 4 and kinds[24:26] == [O.OP_ROTATE] * 2 and kinds[26:28] == [O.OP_FLIP] * 2
    assert kinds[28:31] == [O.OP_COPY, O.OP_COPY, O.OP_PASTE]
    assert kinds[31:35] == [O.OP_COPY_FROM_INPUT, O.OP_RESET_GRID, O.OP_RESIZE_GRID, O.OP_SUBMIT]
    wrapped = [(d >> 16) & 1 for d in ops]
    assert wrapped == [1] * 20 + [0] * 8 + [1] * 6 + [0]  # reset_sel on 0-19 and 28-33 only


def test_oracle_bad_op_and_autoreset():
    env = O.OracleEnv(2, 5, 5, max_trial=1, kind="o2arc")
    a = np.arange(25, dtype=np.int8).reshape(5, 5) % 10
    env.set_tasks([a, a], [a, a])
    env.reset()
    before = env.planes["grid"].copy()
    env.step_point([[0, 0], [0, 0]], [99, -1])
    assert env.status() == O.ST_BAD_OP and np.array_equal(before, env.planes["grid"]) and env.cnt[:, 0].tolist() == [0, 0]
    r, t = env.step_point([[0, 0], [0, 0]], [34, 3])  # env0 submits the correct answer (grid == input == answer)
    assert r.tolist() == [1, 0] and t.tolist() == [1, 0]
    r, t = env.step_point([[1, 1], [1, 1]], [3, 3], flags=O.STEP_AUTORESET)
    assert t.tolist() == [0, 0] and env.cnt[0].tolist() == [0, 0] and env.field("trials_remain")[0, 0] == 1
    assert np.array_equal(env.planes["grid"][0], a)  # env0 was re-initialised, its action ignored
    assert env.planes["grid"][1, 1, 1] == 3
