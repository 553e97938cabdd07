"""arcle_amd.replay.action_convert against the reference's own `action_convert` (tests/o2arc_check.py:21-99), through the
golden (entry -> op, selection) pairs captured by tests/golden/make_golden_research.py.  CPU only."""
import numpy as np

import features as F
from arcle_amd.replay import action_convert


def test_action_convert_matches_reference():
    g = F.golden()
    seen = set()
    for i, trace in enumerate(g["traces"]):
        for t, (name, data) in enumerate(trace):
            op, sel = action_convert((None, name, data, None))
            assert op == int(g["replay_op"][i, t]), (i, t, name, data)
            assert np.array_equal(sel.astype(np.int8), g["replay_sel"][i, t]), (i, t, name, data)
            seen.add(name)
    assert {"Color", "Fill", "Move", "Copy", "Paste", "FloodFill", "ResizeGrid", "Submit"} <= seen
