"""Round-5 GPU tests: the launch that orders itself (ARCLE_STEPX_GROUPED).

Every wave of a 32-env group derives the same permutation of the group from the group's 32 op indices and steps the env its dispatch slot
is dealt — scheduling only.  Checked here: identical rewards, terminated / truncated flags, packed rows, observation rows and every byte
of state against a twin handle created with ARCLE_GROUPED=0, for every ingress form and flag set that has such an instantiation, on the
natural C3 stream and on adversarial op streams (every env the same object operation, no object operation at all, object operations
only in the LAST positions of every group, ...), at several batch sizes incl. ones the grouping does not apply to.  The full-batch
comparison with the oracle at BASELINE sizes is tests/test_round4_hip.py (whole batch since round 5)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _streams(K, n, seed):
    """bench.py's C3 actions, then op streams built to break a slot <-> env permutation: positions of the object ops inside the groups"""
    import bench
    bb, op = bench.make_actions(K, n, seed)
    op = op.copy()
    lng = np.arange(20, 28)
    g = op.reshape(K, n // 32, 32)
    g[1] = 24                                   # every env: the same object operation (L = 32 in every group)
    g[2] = 3                                    # no object operation anywhere (L = 0)
    g[3, :, :] = 5; g[3, :, 31] = 21            # one object op, in the last position
    g[4, :, :16] = 7; g[4, :, 16:] = lng[np.arange(16) % 8]   # the late half all object ops: 16 trades per group
    g[5, :, ::2] = 22; g[5, :, 1::2] = 9        # alternating
    g[6, :, :] = 26; g[6, :, 0] = 34            # all object ops but position 0 (Submit)
    g[7] = np.where(np.random.default_rng(seed).random((n // 32, 32)) < 0.5, 23, 1)
    op[8, : n // 2] = 20                        # half of the batch one object op, the other half the C3 stream
    op[9] = np.where(np.arange(n) % 3 == 0, 40, op[9])  # out-of-range op indices in every group (ARCLE_ST_BAD_OP, skipped steps)
    return bb, op


@pytest.mark.parametrize("variant", ["bbox", "point", "bbox5", "bbox_pack", "bbox5_pack", "bbox_noreset", "point_noreset", "bbox5_noreset"])
@pytest.mark.parametrize("n", [2304, 4096, 8192, 16384, 66560])
def test_grouped_launches_are_scheduling_only(variant, n):
    import torch
    import bench
    from arcle_amd.engine import STEP_PACK_OBS
    K, dev = 16, torch.device("cuda:0")
    bb_np, op_np = _streams(K, n, 77 + n)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    xy = bb[:, :, :2].contiguous()
    act5 = torch.cat([bb, op[:, :, None]], 2).contiguous()
    with _Env(ARCLE_GROUPED=0):
        a = bench.make_batch(dev, n, seed=11)
    with _Env(ARCLE_GROUPED=1, ARCLE_GROUP_MIN=0, ARCLE_GROUP_MAX=10000000):  # (every size here takes the self-ordering launch where it structurally can)
        b = bench.make_batch(dev, n, seed=11)
    FL = a.elide_flag | (0 if variant.endswith("_noreset") else 1)  # (_noreset: ARCVecEnv's default flag set — terminated envs keep being stepped)
    assert b.launch_info("bbox5" if variant.startswith("bbox5") else variant.split("_")[0], FL | (STEP_PACK_OBS if variant.endswith("_pack") else 0))["orders_itself"]
    pa = pb = None
    if variant.endswith("_pack"):
        FL |= STEP_PACK_OBS
        pa, pb = a.set_packed_output(), b.set_packed_output()

    def one(batch, s):
        if variant.startswith("point"):
            return batch.step_point(xy[s], op[s], FL)
        if variant.startswith("bbox5"):
            return batch.step_bbox5(act5[s], FL)
        return batch.step_bbox(bb[s], op[s], FL)
    for rep in range(2):  # (the second pass steps the states the adversarial streams left behind)
        for s in range(K):
            ra, ta = one(a, s)
            ra, ta = ra.clone(), ta.clone()
            rb, tb = one(b, s)
            assert torch.equal(ra, rb) and torch.equal(ta, tb), (variant, n, rep, s)
            if pa is not None:
                assert torch.equal(pa, pb), (variant, n, rep, s)
    torch.cuda.synchronize()
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), (variant, n, k)
    assert torch.equal(a.rec, b.rec) and torch.equal(a.cnt, b.cnt)
    assert a.status() == b.status()


def test_grouped_step_many_and_hints_match_single_steps():
    """arcle_step_many and hinted single steps on a handle whose launches order themselves (the hint is accepted and ignored, step_many
    enqueues plain self-ordering launches) against ungrouped single steps."""
    import torch
    import bench
    n, K, dev = 8192, 12, torch.device("cuda:0")
    bb_np, op_np = _streams(K, n, 5)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    with _Env(ARCLE_GROUPED=0):
        a = bench.make_batch(dev, n, seed=3)
    b, c = bench.make_batch(dev, n, seed=3), bench.make_batch(dev, n, seed=3)
    FL = a.elide_flag | 1
    rm, tm = b.step_many("bbox", bb, op, FL)
    for s in range(K):
        ra, ta = a.step_bbox(bb[s], op[s], FL)
        if s + 1 < K:
            c.hint_next_ops(op[s + 1])
        rc, tc = c.step_bbox(bb[s], op[s], FL)
        assert torch.equal(ra, rm[s]) and torch.equal(ta, tm[s]) and torch.equal(ra, rc) and torch.equal(ta, tc), s
    torch.cuda.synchronize()
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]) and torch.equal(a.planes[k], c.planes[k]), k
    assert torch.equal(a.cnt, b.cnt) and torch.equal(a.cnt, c.cnt) and torch.equal(a.rec, b.rec) and torch.equal(a.rec, c.rec)


def test_grouped_research_step_matches_ungrouped():
    """The research env's flag set (dense reward, TimeLimit, device-drawn augmented tasks, incremental FilterO2ARC rows): the grouped launch
    also counts an env about to be re-initialised as a long wave (its step counter reads limit - 1) — identical to the ungrouped twin."""
    import torch
    import bench
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    n, K, dev = 4096, 30, torch.device("cuda:0")
    bb_np, op_np = _streams(K, n, 99)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np % 35).to(dev)
    kw = dict(device=dev, seed=5, autoreset="resample", augment=("permute", "rot90"), dense_reward=True, max_episode_steps=9)
    with _Env(ARCLE_GROUPED=0):
        va = ARCVecEnv(O2ARCv2Env, n, SyntheticLoader(n_tasks=60, seed=2, max_size=(30, 30)), **kw)
    vb = ARCVecEnv(O2ARCv2Env, n, SyntheticLoader(n_tasks=60, seed=2, max_size=(30, 30)), **kw)
    for v in (va, vb):
        v.reset()
        v.enable_flat_rows(filtered=True)
    for s in range(K):
        _, ra, ta, tra, ia = va.step_bbox(bb[s], op[s])
        _, rb, tb, trb, ib = vb.step_bbox(bb[s], op[s])
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(tra, trb), s
        assert torch.equal(va.rows, vb.rows), s
    for k in va.batch.planes:
        assert torch.equal(va.batch.planes[k], vb.batch.planes[k]), k
    assert torch.equal(va.batch.cnt, vb.batch.cnt)
    va.check_errors(), vb.check_errors()


def test_launch_info_and_autotune_keep_state_and_results():
    """arcle_launch_info reports the plan (self-ordering inside the window, policy letters outside); arcle_autotune times the candidates on
    the caller's own actions, leaves every byte of state untouched, and the steps that follow — now on the tuned plan — still equal the
    untuned twin's."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    for n, want in ((8192, True), (1024, False), (40960, False)):
        b = bench.make_batch(dev, n, seed=2)
        info = b.launch_info("bbox", b.elide_flag | 1)
        assert info["orders_itself"] == want and not info["autotuned"], (n, info)
        assert info["policy"] == {8192: "", 1024: "A", 40960: "B"}[n] and info["waves_per_workgroup"] in (4, 8), (n, info)
        del b
    n, K = 8192, 12
    bb_np, op_np = _streams(K, n, 31)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    a, b = bench.make_batch(dev, n, seed=4), bench.make_batch(dev, n, seed=4)
    FL = a.elide_flag | 1
    for s in range(3):
        a.step_bbox(bb[s], op[s], FL), b.step_bbox(bb[s], op[s], FL)
    before = {k: v.clone() for k, v in b.planes.items()}
    rec, cnt = b.rec.clone(), b.cnt.clone()
    rows = b.autotune("bbox", bb[3:].contiguous(), op[3:].contiguous(), FL)  # (K - 3 consecutive action batches)
    assert len(rows) >= 6 and rows[0]["us_per_launch"] > 0, rows
    assert any(r["orders_itself"] for r in rows) and any(r["policy"] == "B" for r in rows), rows
    for k in before:
        assert torch.equal(before[k], b.planes[k]), k
    assert torch.equal(rec, b.rec) and torch.equal(cnt, b.cnt)
    info = b.launch_info("bbox", FL)
    assert info["autotuned"] and info["orders_itself"] == rows[0]["orders_itself"] and info["policy"] == rows[0]["policy"], (info, rows[0])
    for s in range(3, K):
        ra, ta = a.step_bbox(bb[s], op[s], FL)
        rb, tb = b.step_bbox(bb[s], op[s], FL)
        assert torch.equal(ra, rb) and torch.equal(ta, tb), s
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), k
    assert a.status() == b.status()


def test_host_resident_records_keep_per_env_loads():
    """A self-ordering launch reads the actions of 32 envs per wave; records in pinned host memory (ARCVecEnv.step_bbox5 accepts them) must
    keep the scalar per-env loads — same results either way."""
    import torch
    import bench
    n, K, dev = 4096, 10, torch.device("cuda:0")
    bb_np, op_np = _streams(K, n, 8)
    act5 = torch.cat([torch.from_numpy(bb_np), torch.from_numpy(op_np)[:, :, None]], 2).contiguous()
    host, devc = act5.pin_memory(), act5.to(dev)
    a, b = bench.make_batch(dev, n, seed=6), bench.make_batch(dev, n, seed=6)
    FL = a.elide_flag | 1
    for s in range(K):
        ra, ta = a.step_bbox5(devc[s], FL)
        rb, tb = b.step_bbox5(host[s], FL)
        assert torch.equal(ra, rb) and torch.equal(ta, tb), s
    torch.cuda.synchronize()
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), k


@pytest.mark.parametrize("n", [192, 8192])
def test_rollout_emits_the_packed_row_of_every_step(n):
    """arcle_rollout_bbox with ARCLE_STEP_PACK_OBS (round 5): T steps in one launch, state resident in registers, AND the packed
    observation row of every step — equal to the rows T single steps pack, reward / terminated / final state included."""
    import torch
    import bench
    from arcle_amd.engine import STEP_PACK_OBS
    T, dev = 24, torch.device("cuda:0")
    bb_np, op_np = _streams(T, n, 13) if n % 32 == 0 and n >= 2304 else bench.make_actions(T, n, 13)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np % 35).to(dev)
    a, b = bench.make_batch(dev, n, seed=21), bench.make_batch(dev, n, seed=21)
    FL = a.elide_flag | 1
    R = a.packed_obs_size()
    rows = torch.full((T, n, R), 0x55, dtype=torch.uint8, device=dev)
    rr, tt = b.rollout(bb, op, FL, packed=rows)
    pa = a.set_packed_output()
    for s in range(T):
        ra, ta = a.step_bbox(bb[s], op[s], FL | STEP_PACK_OBS)
        assert torch.equal(ra, rr[s]) and torch.equal(ta, tt[s]), s
        assert torch.equal(pa, rows[s]), f"packed rows of step {s} differ"
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), k
    assert torch.equal(a.rec, b.rec) and torch.equal(a.cnt, b.cnt)


def test_vec_env_autotune_keeps_results():
    """ARCVecEnv.autotune on the caller's own action batches: steps afterwards equal an untuned twin's; envs with side-state flag sets refuse."""
    import torch
    import bench
    from arcle_amd.envs import ARCVecEnv, O2ARCv2Env
    from arcle_amd.loaders import SyntheticLoader
    n, K, dev = 4096, 8, torch.device("cuda:0")
    bb_np, op_np = bench.make_actions(K, n, 3)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    mk = lambda **kw: ARCVecEnv(O2ARCv2Env, n, SyntheticLoader(n_tasks=30, seed=2, max_size=(30, 30)), device=dev, seed=4, autoreset=True, **kw)
    va, vb = mk(), mk()
    va.reset(), vb.reset()
    plans = vb.autotune(bb, op)
    assert len(plans) >= 6 and plans[0]["us_per_launch"] <= plans[-1]["us_per_launch"]
    for s in range(K):
        oa, ra, ta, _, _ = va.step_bbox(bb[s], op[s])
        ob, rb, tb, _, _ = vb.step_bbox(bb[s], op[s])
        assert torch.equal(ra, rb) and torch.equal(ta, tb) and torch.equal(oa["grid"], ob["grid"]), s
    with pytest.raises(NotImplementedError):
        mk(dense_reward=True).autotune(bb, op)


def test_autotune_on_an_env_kind_without_selected_plane():
    """ARCEnv handles (no `selected` plane: the launcher adds the vacuous elision flag itself) keep the tuned plan too, and step the same."""
    import torch
    import bench
    n, K, dev = 4096, 8, torch.device("cuda:0")
    bb_np, op_np = bench.make_actions_c5(K, n, 3)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    a, b = bench.make_batch(dev, n, seed=5, kind="arc"), bench.make_batch(dev, n, seed=5, kind="arc")
    rows = b.autotune("bbox", bb, op, 1)
    assert rows and b.launch_info("bbox", 1)["autotuned"] and not a.launch_info("bbox", 1)["autotuned"]
    for s in range(K):
        ra, ta = a.step_bbox(bb[s], op[s], 1)
        rb, tb = b.step_bbox(bb[s], op[s], 1)
        assert torch.equal(ra, rb) and torch.equal(ta, tb), s
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), k


@pytest.mark.parametrize("form", ["mask", "bits"])
@pytest.mark.parametrize("n", [2304, 8192])
def test_grouped_launches_with_mask_ingress(form, n):
    """Full int8 masks and bit-packed masks order themselves too (the mask of the slot's env is fetched once the env is known): identical to
    the plain launches, adversarial op streams included."""
    import torch
    import bench
    K, dev = 10, torch.device("cuda:0")
    bb_np, op_np = _streams(K, n, 5 + n)
    op = torch.from_numpy(op_np).to(dev)
    bb = torch.from_numpy(bb_np).to(dev)
    # rectangle masks of the bbox tuples (BBoxWrapper.action, bbox.py:22-30), a few cells set to 2 / -1 so that they are not plain wrapper masks
    r = torch.arange(30, device=dev)
    x1, x2 = torch.minimum(bb[..., 0], bb[..., 2]), torch.maximum(bb[..., 0], bb[..., 2])
    y1, y2 = torch.minimum(bb[..., 1], bb[..., 3]), torch.maximum(bb[..., 1], bb[..., 3])
    rows = (r[None, None, :] >= x1[..., None]) & (r[None, None, :] <= x2[..., None])
    cols = (r[None, None, :] >= y1[..., None]) & (r[None, None, :] <= y2[..., None])
    masks = (rows[..., :, None] & cols[..., None, :]).to(torch.int8)
    if form == "mask":
        masks[:, ::7, 3, 4] = 2
        masks[:, ::11, 0, 0] = -1
    with _Env(ARCLE_GROUPED=0):
        a = bench.make_batch(dev, n, seed=9)
    b = bench.make_batch(dev, n, seed=9)
    FL = a.elide_flag | 1
    assert b.launch_info(form, FL)["orders_itself"] and not a.launch_info(form, FL)["orders_itself"]
    for s in range(K):
        if form == "mask":
            ra, ta = a.step_mask(masks[s], op[s], FL)
            ra, ta = ra.clone(), ta.clone()
            rb, tb = b.step_mask(masks[s], op[s], FL)
        else:
            bits = a.pack_mask_bits(masks[s])
            ra, ta = a.step_bits(bits, op[s], FL)
            ra, ta = ra.clone(), ta.clone()
            rb, tb = b.step_bits(bits, op[s], FL)
        assert torch.equal(ra, rb) and torch.equal(ta, tb), (form, n, s)
    torch.cuda.synchronize()
    for k in a.planes:
        assert torch.equal(a.planes[k], b.planes[k]), (form, n, k)
    assert torch.equal(a.rec, b.rec) and torch.equal(a.cnt, b.cnt) and a.status() == b.status()


def test_autotune_leaves_packed_rows_and_status_alone():
    """the 70-160 timed launches of arcle_autotune with ARCLE_STEP_PACK_OBS used to overwrite the caller's packed-row buffer
    with tuning-run observations, and the sticky status word could pick up their ST_* bits.  Both are part of "the handle as found"."""
    import torch
    import bench
    from arcle_amd.engine import STEP_PACK_OBS
    dev, n, K = torch.device("cuda:0"), 4096, 12
    bb_np, op_np = _streams(K, n, 77)
    op_np[:, ::97] = 63  # (op indices beyond the table: every timed launch raises ARCLE_ST_BAD_OP)
    bb, op = torch.from_numpy(bb_np).to(dev), torch.from_numpy(op_np).to(dev)
    b = bench.make_batch(dev, n, seed=9)
    FL = b.elide_flag | 1 | STEP_PACK_OBS
    packed = b.set_packed_output()
    good = torch.from_numpy(np.where(op_np[0] == 63, 0, op_np[0])).to(dev)
    b.step_bbox(bb[0], good, FL)
    assert b.status(clear=True) == 0
    rows_before = packed.clone()
    rows = b.autotune("bbox", bb, op, FL)
    assert rows
    assert torch.equal(packed, rows_before), "the tuning launches wrote into the caller's packed rows"
    assert b.status(clear=False) == 0, "the tuning launches left their status bits behind"
    b.step_bbox(bb[1], op[1], FL)  # ... and the buffer is installed again afterwards
    assert not torch.equal(packed, rows_before) and b.status() != 0


def test_launch_info_reports_the_big_grid_workgroup():
    """one workgroup per env beyond 1024 cells: two chunks per thread in the LEAN step launches (64 threads = one wavefront at 40 x 40), the
    generic kernel's size when a flag outside their set is on"""
    import torch
    import bench
    from arcle_amd.engine import STEP_FLAT_OBS
    dev = torch.device("cuda:0")
    for (H, W), lean_waves, generic_waves in (((40, 40), 1, 2), ((64, 64), 2, 4), ((127, 127), 8, 8)):
        b = bench.make_batch(dev, 64, 3, "o2arc", H, W)
        assert b.launch_info("bbox", b.elide_flag | 1)["waves_per_workgroup"] == lean_waves, (H, W)
        assert b.launch_info("bbox", b.elide_flag | 1 | STEP_FLAT_OBS)["waves_per_workgroup"] == generic_waves, (H, W)
