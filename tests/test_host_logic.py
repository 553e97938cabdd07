"""Host-side logic that needs no GPU: op descriptors (names, wrappers, table validation), loaders (ARC JSON layout,
MiniARC quirks, Loader.pick), wrappers' mask arithmetic, the gymnasium-free spaces."""
import json
import os

import numpy as np
import pytest

from arcle_amd import actions as A
from arcle_amd import loaders, spaces, wrappers
from arcle_amd.envs import ARCEnv, O2ARCv2Env, RawARCEnv


def names(ops):
    return ["".join(map(str.capitalize, op.__name__.split("_"))) for op in ops]


def test_op_names_match_reference_capitalisation():
    # base.py:66 applied to the reference's __name__s
    n = names(O2ARCv2Env.default_operations())
    assert n[0] == "Color0" and n[13] == "Floodfill3" and n[20:24] == ["MoveU", "MoveD", "MoveR", "MoveL"]
    assert n[24:28] == ["Rotate90", "Rotate270", "FlipH", "FlipV"]
    assert n[28:] == ["CopyI", "CopyO", "Paste", "CopyFromInput", "ResetGrid", "ResizeGrid", "Submit"]
    assert names(RawARCEnv.default_operations())[10:] == ["ResizeToAnswer", "Submit"]
    assert len(ARCEnv.default_operations()) == 27 and names(ARCEnv.default_operations())[-1] == "Submit"


def test_descriptor_encoding_and_wrappers():
    op = A.reset_sel(A.gen_flood_fill(7))
    assert op.desc == A.OP_FLOODFILL | (7 << 8) | (A.OPF_RESET_SEL << 16) and op.__name__ == "FloodFill7"
    both = A.keep_sel(A.reset_sel(A.gen_color(3)))
    assert (both.desc >> 16) == 3 and both.__name__ == "Color3"
    assert A.gen_flip("D1").arg == 3 and A.gen_copy("O").arg == 1 and A.gen_paste(True).arg == 1
    for bad in (lambda: A.gen_move(4), lambda: A.gen_rotate(0), lambda: A.gen_rotate(4), lambda: A.gen_flip("X"),
                lambda: A.gen_copy("Z")):
        with pytest.raises(AssertionError):  # the reference asserts too (object.py:175,226,261,289)
            bad()
    # an arbitrary callable becomes a device no-op slot that the env applies on the host (base.py:140-142)
    table = [A.gen_color(1), lambda s, a: None]
    assert A.table_descs(table) == [A.gen_color(1).desc, A.OP_HOST] and A.host_slots(table) == [1]
    with pytest.raises(TypeError):
        A.table_descs([A.gen_color(1), 42])
    with pytest.raises(TypeError):
        A.gen_color(1)({}, {})  # descriptors are not host callables


def test_loader_pick_semantics(tmp_path):
    class Two(loaders.Loader):
        def get_path(self, **kw):
            return ["a", "b"]

        def parse(self, **kw):
            g = lambda v: np.full((2, 2), v, np.int8)  # noqa: E731
            return [([g(1)], [g(2)], [g(3)], [g(4)], {"id": "a"}), ([g(5)], [g(6)], [g(7)], [g(8)], {"id": "b"})]

    ld = Two()
    assert ld.pick(1)[4]["id"] == "b" and ld.pick(data_index=0)[0][0][0, 0] == 1
    with pytest.raises(AssertionError):
        ld.pick(2)
    assert Two(rng=np.random.default_rng(0)).pick()[4]["id"] in ("a", "b")


def test_arc_and_miniarc_json_layout(tmp_path):
    root = tmp_path / "data"
    (root / "training").mkdir(parents=True)
    (root / "evaluation").mkdir()
    task = {"train": [{"input": [[1, 2], [3, 4]], "output": [[4, 3], [2, 1]]}], "test": [{"input": [[0]], "output": [[9]]}]}
    (root / "training" / "abc123.json").write_text(json.dumps(task))
    (root / "training" / "000aaa.json").write_text(json.dumps(task))
    ld = loaders.ARCLoader(train=True, root=str(root))
    assert [d[4]["id"] for d in ld.data] == ["000aaa", "abc123"]  # sorted paths, loader.py:86
    ti, to, ei, eo, _ = ld.data[0]
    assert ti[0].dtype == np.int8 and to[0].tolist() == [[4, 3], [2, 1]] and eo[0].tolist() == [[9]]
    assert loaders.ARCLoader(train=False, root=str(root)).data == []
    mini = tmp_path / "mini"
    mini.mkdir()
    (mini / "some_description_l6abcd.json").write_text('{"train": [{"input": [[null, 1]], "output": [[1, 1]]}], "test": [{"input": [[2]], "output": [[2]]}]}')
    m = loaders.MiniARCLoader(root=str(mini))
    assert m.data[0][4] == {"id": "l6abcd", "description": "some description"}
    assert m.data[0][0][0].tolist() == [[0, 1]]  # null -> "0" -> int8 0, loader.py:139


def test_synthetic_loader_is_deterministic_and_arc_shaped():
    a = loaders.SyntheticLoader(n_tasks=5, max_size=(9, 7), seed=4)
    b = loaders.SyntheticLoader(n_tasks=5, max_size=(9, 7), seed=4)
    assert len(a.data) == 5
    for (ti, to, ei, eo, d), (ti2, *_rest) in zip(a.data, b.data):
        assert len(ti) == len(to) == 2 and len(ei) == len(eo) == 1 and d["id"].startswith("synthetic")
        assert all(g.dtype == np.int8 and g.shape[0] <= 9 and g.shape[1] <= 7 and g.min() >= 0 and g.max() <= 9 for g in ti + to + ei + eo)
        assert np.array_equal(ti[0], ti2[0])


class _FakeEnv(spaces.Env):
    H, W = 5, 6
    operations = list(range(12))

    def step(self, action):
        return action


def test_wrappers_build_the_reference_masks():
    bw = wrappers.BBoxWrapper(_FakeEnv())
    a = bw.action((3, 4, 1, 2, 7))  # unsorted corners (bbox.py:24-29)
    m = np.zeros((5, 6), np.int8)
    m[1:4, 2:5] = 1
    assert a["operation"] == 7 and a["selection"].dtype == np.int8 and np.array_equal(a["selection"], m)
    assert bw.step((0, 0, 0, 0, 1))["selection"].sum() == 1
    assert [s.n for s in bw.action_space.spaces] == [5, 6, 5, 6, 12]
    pw = wrappers.PointWrapper(_FakeEnv())
    a = pw.action((4, 5, 3))
    assert a["selection"][4, 5] == 1 and a["selection"].sum() == 1 and [s.n for s in pw.action_space.spaces] == [5, 6, 12]


def test_spaces_fallback_sampling():
    if spaces.HAVE_GYMNASIUM:
        pytest.skip("real gymnasium present")
    d = spaces.Discrete(5)
    d.seed(0)
    assert all(0 <= d.sample() < 5 for _ in range(20))
    assert d.sample(mask=np.array([0, 0, 1, 0, 0])) == 2
    t = spaces.Tuple((spaces.Discrete(3), spaces.Discrete(4)))
    assert len(t.sample()) == 2
    b = spaces.Box(0, 1, (3, 3), dtype=np.int8)
    assert b.sample().shape == (3, 3) and b.sample().dtype == np.int8
    dd = spaces.Dict({"selection": b, "operation": d})
    assert set(dd.sample()) == {"selection", "operation"} and dd["operation"].n == 5


def test_lazy_info_dict_semantics():
    """ARCVecEnv's info: entries registered lazily behave like ordinary dict entries and are computed once, on first access."""
    from arcle_amd.envs.vec import _LazyInfo
    calls = []
    d = _LazyInfo({"steps": 3})
    d.lazy("task_index", lambda: calls.append(1) or 7)
    assert "task_index" in d and "steps" in d and "nope" not in d and len(d) == 2 and calls == []
    assert d["steps"] == 3 and calls == []          # eager entries never trigger the thunk
    assert d["task_index"] == 7 and d["task_index"] == 7 and calls == [1]
    e = _LazyInfo({"a": 1})
    e.lazy("b", lambda: 2)
    assert e.get("b") == 2 and e.get("c", 5) == 5
    f = _LazyInfo({"a": 1})
    f.lazy("b", lambda: 2)
    assert sorted(f.keys()) == ["a", "b"] and dict(f.items()) == {"a": 1, "b": 2} and sorted(f) == ["a", "b"]
    g = _LazyInfo({"a": 1})
    g.lazy("b", lambda: 2)
    g["b"] = 9  # an explicit assignment wins over the pending thunk
    assert g["b"] == 9 and dict(g.items()) == {"a": 1, "b": 9}
    try:
        g["zzz"]
        raise AssertionError("KeyError expected")
    except KeyError:
        pass


def test_table_probe_falls_back_to_constructing_the_class():
    """ARCVecEnv asks the env CLASS for its op table: create_operations() on a bare instance first (agents/env.py:23-28 style overrides
    need nothing else), the real constructor when the override reads instance state set up in __init__."""
    from arcle_amd.envs.vec import _table_of
    from arcle_amd.loaders import SyntheticLoader

    class NeedsInit(O2ARCv2Env):
        def __init__(self, *a, **k):
            self.extra_colour = 7          # (set BEFORE the base constructor builds the table)
            super().__init__(*a, **k)

        def create_operations(self):
            ops = super().create_operations()
            ops[0] = A.reset_sel(A.gen_color(self.extra_colour))
            return ops
    kw = dict(data_loader=SyntheticLoader(n_tasks=2, seed=0, max_size=(5, 5)), max_grid_size=(5, 5), colors=10, max_trial=-1, device=None)
    table = _table_of(NeedsInit, **kw)
    assert len(table) == 35 and table[0].arg == 7 and table[0].__name__ == "Color7"
    assert [op.desc for op in _table_of(O2ARCv2Env, **kw)] == [op.desc for op in O2ARCv2Env.default_operations()]


def test_lazy_info_dict_semantics():
    from arcle_amd.envs.vec import _LazyInfo
    calls = []
    d = _LazyInfo({"a": 1})
    d.lazy("b", lambda: calls.append("b") or 2)
    d.lazy("c", lambda: calls.append("c") or 3)
    assert "b" in d and len(d) == 3 and calls == []
    c = d.copy()
    assert c == {"a": 1, "b": 2, "c": 3} and type(c) is dict and sorted(calls) == ["b", "c"]
    d2 = _LazyInfo({})
    d2.lazy("x", lambda: 5)
    assert d2.pop("x") == 5 and "x" not in d2 and d2.pop("x", None) is None
    d3 = _LazyInfo({})
    d3.lazy("y", lambda: 9)
    assert d3.setdefault("y", 0) == 9 and d3.setdefault("z", 4) == 4


def test_dense_reward_formula_and_no_action_marker():
    """ARCVecEnv._dense: sparse * 100 - 1 + correct / total (agents/env.py:44-58); the pair (0, 0) — a step that executed no action —
    is reward 0."""
    import torch
    from arcle_amd.envs import ARCVecEnv
    r = ARCVecEnv._dense(torch.tensor([0, 1, 0, 0]), torch.tensor([[3, 4], [9, 9], [0, 0], [0, 5]]))
    assert r.dtype == torch.float32 and r.tolist() == [-0.25, 100.0, 0.0, -1.0]


def test_augmentation_draw_mirror_matches_the_scalar_mirror():
    from arcle_amd import sampling as S
    gids, eps = [0, 5, 123456789, 2**40 + 3], [0, 1, 7, 300]
    for flags in (0, 1, 2, 3):
        k, perm = S.draw_aug_batch(99, gids, eps, flags)
        for i, (g, e) in enumerate(zip(gids, eps)):
            _, _, kk, pp = S.draw_task(99, g, e, [2, 3, 4], flags)
            assert kk == int(k[i]) and pp == perm[i].tolist()


@pytest.mark.parametrize("size", [(128, 4), (4, 128), (200, 200)])
def test_grid_side_beyond_int8_is_refused_by_name(size):
    """The reference takes any max_grid_size (base.py:37-49) but stores the dims as int8 (base.py:162-166): sides up to 127 are served (planes of
    more than 1024 cells by the workgroup-per-env kernels, tests/test_big_*.py); a longer side is refused by every constructor with a
    ValueError that names the limit — before it touches a device (this runs on CPU)."""
    from arcle_amd.engine import EnvBatch
    from arcle_amd.envs import ARCEnv, ARCVecEnv, O2ARCv2Env, RawARCEnv
    from arcle_amd.loaders import SyntheticLoader
    loader = SyntheticLoader(n_tasks=2, seed=0, max_size=(5, 5))
    for make in (lambda: EnvBatch(4, *size), lambda: O2ARCv2Env(data_loader=loader, max_grid_size=size),
                 lambda: ARCEnv(data_loader=loader, max_grid_size=size), lambda: RawARCEnv(data_loader=loader, max_grid_size=size),
                 lambda: ARCVecEnv(O2ARCv2Env, 4, loader, max_grid_size=size)):
        with pytest.raises(ValueError, match=r"H, W <= 127"):
            make()


def test_largest_supported_grids_pass_the_size_check():
    from arcle_amd.engine import check_grid_size
    from arcle_amd.engine import is_big_grid
    for size in ((32, 32), (30, 30), (8, 127), (127, 8), (1, 127), (1, 1)):
        check_grid_size(*size)
        assert not is_big_grid(*size)
    for size in ((33, 32), (40, 40), (127, 127), (9, 127)):
        check_grid_size(*size)
        assert is_big_grid(*size)
    with pytest.raises(ValueError):
        check_grid_size(0, 5)
