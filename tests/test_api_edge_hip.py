"""Python-level edges of the single-env boundary: arcle_amd's classes replay the script tests/golden/make_golden_api.py ran on the
reference's own classes and must return the same thing from every call (tests/apiedge.py; o2arcenv.py:121-151, arcenv.py:51-76,
base.py:69-118,172-183)."""
import copy
import json
import os

import numpy as np
import pytest

import apiedge as AE

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    with open(os.path.join(GOLDEN, "api_edge_script.json")) as f:
        script = json.load(f)
    return script, np.load(os.path.join(GOLDEN, "api_edge.npz"))


def test_fixture_covers_the_edges_it_names():
    script, z = _load()
    kinds, dtypes, layouts, op_as, neg = set(), set(), set(), set(), 0
    for i, sc in enumerate(script["scenarios"]):
        assert z[f"s{i}_rows"].shape[0] == len(sc["calls"]) == z[f"s{i}_scal"].shape[0]
        assert z[f"s{i}_sels"].shape[1:] == (sc["H"], sc["W"])
        for c in sc["calls"]:
            kinds.add((c["k"], c.get("on")))
            dtypes.add(c.get("dtype", "int8"))
            layouts.add(c.get("layout", "c"))
            op_as.add(c.get("op_as", "int"))
            neg += c.get("op", 0) < 0
    assert {("reset", None), ("step", None), ("transition", "deepcopy"), ("transition", "live"), ("submit", "deepcopy"),
            ("submit", "live")} <= kinds
    assert {"bool", "uint8", "int8", "int16", "int32", "int64", "float32", "float64"} <= dtypes
    assert {"c", "f", "strided", "reversed", "transposed", "readonly"} <= layouts
    assert {"int", "int64", "int8", "int32", "float", "arr0", "arr1"} <= op_as
    assert neg >= 100
    # the verdict's case: Submit by index -1 on a solved grid -> reward 0, terminated (o2arcenv.py:121-136)
    s0 = z["s0_scal"]
    assert script["scenarios"][0]["calls"][1] == {"k": "step", "op": -1, "sel": 0}
    assert s0[1, AE.COLS.index("reward")] == 0 and s0[1, AE.COLS.index("terminated")] == 1 and s0[1, AE.COLS.index("last_action_op")] == -1
    assert s0[4, AE.COLS.index("reward")] == 1  # ... and reward 1 by its positive index


def _make_env(cls, tasks, H, W, max_trial):
    from arcle_amd import envs, loaders
    klass = {"o2arc": envs.O2ARCv2Env, "arc": envs.ARCEnv, "raw": envs.RawARCEnv}[cls]
    return klass(data_loader=AE.make_loader(loaders.Loader, tasks), max_grid_size=(H, W), colors=10, max_trial=max_trial)


@pytest.mark.gpu
def test_api_edge_fixture_replays_on_the_hip_classes():
    import warnings
    script, z = _load()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = AE.run_script(script, _make_env, [z[f"s{i}_sels"] for i in range(len(script["scenarios"]))])
    errs = AE.compare(script, got, z)
    assert not errs, f"{len(errs)} mismatches\n" + "\n".join(errs[:25])


@pytest.mark.gpu
def test_submit_by_negative_index_on_a_solved_grid():
    script, _ = _load()
    env = _make_env("o2arc", script["tasks"], 30, 30, -1)
    env.reset(options={"prob_index": 0, "subprob_index": 0})  # (answer == input: solved from the start)
    sel = np.zeros((30, 30), np.int8)
    _, r, term, _, _ = env.step({"selection": sel, "operation": -1})
    assert (r, term, env.last_action_op, env.submit_count) == (0, True, -1, 1)
    env.reset(options={"prob_index": 0, "subprob_index": 0})
    _, r, term, _, _ = env.step({"selection": sel, "operation": 34})
    assert (r, term, env.last_action_op) == (1, True, 34)


@pytest.mark.gpu
def test_masks_not_representable_as_int8_are_refused_by_name():
    script, _ = _load()
    env = _make_env("o2arc", script["tasks"], 30, 30, -1)
    env.reset(options={"prob_index": 1, "subprob_index": 0})
    before = AE.flatten(env.current_state)
    for bad in (np.full((30, 30), 200, np.uint8), np.full((30, 30), 0.5), np.full((30, 30), 256, np.int64), np.full((30, 30), np.nan)):
        with pytest.raises(ValueError, match="representable"):
            env.step({"selection": bad, "operation": 3})
        with pytest.raises(ValueError, match="representable"):
            env.transition(copy.deepcopy(env.current_state), {"selection": bad, "operation": 3})
    assert np.array_equal(before, AE.flatten(env.current_state)) and env.action_steps == 0


@pytest.mark.gpu
def test_transition_on_the_live_state_then_step_then_reset_with_an_overriding_subclass():
    """the dict step() hands out must survive transition(current_state) (keys rebound or not), a subclass whose
    transition() calls super(), and the next reset() must not return the previous episode's state."""
    from arcle_amd import envs, loaders
    script, _ = _load()
    rng = np.random.default_rng(5)

    class Sub(envs.O2ARCv2Env):
        calls = 0

        def transition(self, state, action):
            Sub.calls += 1
            super().transition(state, action)

    def actions(n):
        out = []
        for _ in range(n):
            m = np.zeros((12, 12), np.int8)
            x, y = rng.integers(0, 6, 2)
            m[x:x + rng.integers(1, 4), y:y + rng.integers(1, 4)] = 1
            out.append({"selection": m, "operation": int(rng.choice([1, 4, 20, 21, 24, 26, 28, 30, 31, 33, 12]))})
        return out

    acts = actions(40)
    mk = lambda k: k(data_loader=AE.make_loader(loaders.Loader, script["tasks"]), max_grid_size=(12, 12), max_trial=3)  # noqa: E731
    plain, sub, viaT = mk(envs.O2ARCv2Env), mk(Sub), mk(envs.O2ARCv2Env)
    for ep in range(2):
        opt = {"prob_index": 1 + ep, "subprob_index": 0}
        o0, _ = plain.reset(options=opt)
        o1, _ = sub.reset(options=opt)
        o2, _ = viaT.reset(options=opt)
        assert np.array_equal(AE.flatten(o0), AE.flatten(o1)) and np.array_equal(AE.flatten(o0), AE.flatten(o2)), "reset returned a stale state"
        kept = o2
        for i, a in enumerate(acts):
            s0, r0, t0, _, _ = plain.step(a)
            s1, r1, t1, _, _ = sub.step(a)
            # third env: the op through transition() on the live dict, every 4th time after rebinding a key the reference-op way
            if i % 4 == 0:
                viaT.current_state["grid"] = viaT.current_state["grid"].copy()
            viaT.transition(viaT.current_state, a)
            assert viaT.current_state is kept
            f0 = AE.flatten(s0)
            assert np.array_equal(f0, AE.flatten(s1)) and (r0, t0) == (r1, t1), f"subclass diverged at step {i}"
            assert np.array_equal(f0, AE.flatten(viaT.current_state)), f"live transition diverged at step {i}"
            if i % 7 == 6:  # the resident device state followed the live transitions: a no-op step returns the same state
                s2, _, _, _, _ = viaT.step({"selection": np.zeros((12, 12), np.int8), "operation": 20})
                s0b, _, _, _, _ = plain.step({"selection": np.zeros((12, 12), np.int8), "operation": 20})
                sub.step({"selection": np.zeros((12, 12), np.int8), "operation": 20})
                assert s2 is kept and np.array_equal(AE.flatten(s2), AE.flatten(s0b)), f"step after live transitions diverged at {i}"
        final = AE.flatten(kept).copy()
        if ep == 0:
            prev = (kept, final)
    assert Sub.calls == 2 * (len(acts) + len(acts) // 7)
    assert np.array_equal(AE.flatten(prev[0]), prev[1]), "reset() overwrote the observation kept from the previous episode"


@pytest.mark.gpu
def test_tail_signal_cannot_be_satisfied_by_a_stale_sequence_number():
    """one 1..255 counter arms the step() tail and the transition() tail; 254 launches on one buffer used to bring the
    counter back to the number the other buffer's stale tail still held."""
    script, _ = _load()
    env = _make_env("o2arc", script["tasks"], 30, 30, -1)
    ref = _make_env("o2arc", script["tasks"], 30, 30, -1)
    for e in (env, ref):
        e.reset(options={"prob_index": 2, "subprob_index": 0})
    sel = np.zeros((30, 30), np.int8)
    sel[1:3, 0:2] = 1
    for rounds in range(3):
        a = {"selection": sel, "operation": 3 + rounds}
        side = copy.deepcopy(env.current_state)
        env.transition(side, a)          # arms the transition() tail ...
        st, _, _, _, _ = ref.step(a)
        assert np.array_equal(AE.flatten(side), AE.flatten(st))
        env.step(a)
        for i in range(254):             # ... and 254 + 1 launches on the step() tail bring the counter back to it
            b = {"selection": sel, "operation": 20 + (i & 3)}
            env.step(b)
            ref.step(b)
        assert np.array_equal(AE.flatten(env.current_state), AE.flatten(ref.current_state))
