#!/usr/bin/env python3
"""Differential fuzz: oracle/arcle_oracle.c  vs  the UNMODIFIED reference in /root/reference.

Build-container only (needs /root/reference; the gymnasium/pygame stubs are in oracle/stubs).
This is pin (1) of the oracle (see arcle_oracle.c header): every op of every env variant is
stepped in lock-step in the reference (one Python env) and in the C restatement, and EVERY state
field, the reward, `terminated`, `steps` and `submit_count` are compared after every step.

    python oracle/diff_vs_reference.py [--traces 400] [--steps 120] [--seed 1] [--big]

--big: max_grid_size beyond 1024 cells (40x40 ... 127x127; the grids libarcle_hip.so steps with one workgroup per env,
arcle_big.hip).  The reference's flood fill recurses once per cell (color.py:16-28), so the run raises Python's recursion
limit and runs on a thread with a large stack — harness settings, the reference itself is untouched.

Exit code 0 iff there was no mismatch.  Where the reference raises (IndexError for a bad op,
ValueError/OverflowError inside Rotate out of its domain) the oracle must have flagged the step
and left the state untouched; the reference env is then re-synchronised from the oracle.
"""
import argparse
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from oracle import refdriver as RD  # noqa: E402


def compare(ref_env, orc, tag):
    ref = RD.flatten_state(ref_env.current_state)
    bad = []
    for k, v in ref.items():
        if k in orc.planes:
            mine = orc.planes[k][0]
        else:
            mine = orc.field(k)[0]
        # keep_sel (object.py:38) stores the caller's mask object itself, so a bool mask leaves a bool
        # `selected` in the reference; values are compared, the dtype quirk is not reproduced.
        if v.dtype != np.int8 and not (k == "selected" and v.dtype == np.bool_):
            bad.append(f"{k}: reference dtype {v.dtype}")
        if v.shape != mine.shape or not np.array_equal(v.astype(np.int64), mine.astype(np.int64)):
            bad.append(f"{k}: ref={v.tolist() if v.size < 8 else '...'} oracle={mine.tolist() if mine.size < 8 else '...'}")
    if bad:
        print(f"MISMATCH {tag}: " + "; ".join(bad))
    return not bad


def sync_reference_from_oracle(ref_env, orc):
    d = orc.state_dict(0)
    st = ref_env.current_state
    for k, v in d.items():
        if k == "object_states":
            for k2, v2 in v.items():
                st["object_states"][k2] = v2
        elif k in st:
            st[k] = v


BIG_SIZES = [(40, 40), (33, 48), (48, 33), (64, 64), (127, 127), (100, 20), (20, 100), (45, 45)]


def run_trace(tid, seed, steps, verbose=False, big=False):
    rng = RD.SplitMix64(seed * 1000003 + tid)
    variant = ["o2arc", "o2arc", "o2arc", "o2arc_crop", "o2arc_exotic", "arc", "raw"][rng.below(7)]
    size = (BIG_SIZES if big else [(5, 5), (10, 10), (12, 12), (30, 30), (30, 30), (7, 12), (12, 7), (3, 3)])[rng.below(8)]
    H, W = size
    max_trial = [-1, -1, 3, 127, 1][rng.below(5)]
    use_bool = rng.chance(1, 3) and variant != "o2arc_exotic"  # keep_sel + bool mask: see compare()
    weird = rng.chance(1, 4) and not use_bool
    runaway = rng.chance(1, 6)
    task = RD.random_task(rng, H, W)
    kind, table = RD.variant_table(variant)
    ref = RD.make_reference_env(variant, H, W, max_trial, task)
    orc = O.OracleEnv(1, H, W, max_trial, kind, table)
    orc.set_tasks([task[0]], [task[1]])
    orc.reset()
    tag0 = f"trace {tid} {variant} {H}x{W} mt={max_trial} bool={use_bool}"
    if not compare(ref, orc, tag0 + " after reset"):
        return 1, 0
    n_ops = len(table)
    nerr = 0
    nexc = 0
    run_dir = None
    for t in range(steps):
        sk, payload, mask = RD.random_selection(rng, H, W, weird)
        op = RD.pick_op(rng, n_ops, variant)
        if rng.chance(1, 200):
            op = n_ops + rng.below(3)  # out of range
        if runaway and variant.startswith("o2arc"):
            # drive an object far off the grid to exercise the int8 wrap of object_pos, then transform it
            period = [40, 90, 160][tid % 3]
            if t % period == 0:
                run_dir = 20 + rng.below(4)
            elif t % period < period - 6:
                op, sk, payload, mask = run_dir, "mask", None, np.zeros((H, W), np.int8)
                if variant == "o2arc_exotic" and op == 20:
                    op = 21
            elif rng.chance(1, 2):
                op, sk, payload, mask = 24 + rng.below(4), "mask", None, np.zeros((H, W), np.int8)
        sel_ref = mask.astype(bool) if use_bool else mask
        action = {"selection": sel_ref.copy(), "operation": op}
        raised = None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                _, r_ref, term_ref, trunc_ref, info = ref.step(action)
            except (IndexError, TypeError, ValueError, OverflowError) as e:  # noqa: PERF203
                raised = e
        if sk == "bbox":
            r, term = orc.step_bbox([payload], [op])
        elif sk == "point":
            r, term = orc.step_point([payload], [op])
        else:
            r, term = orc.step_mask(mask[None], [op])
        st = orc.status()
        tag = f"{tag0} step {t} op {op} sel={sk}{payload if payload else ''}"
        if raised is not None:
            nexc += 1
            if st == 0:
                print(f"MISMATCH {tag}: reference raised {raised!r} but oracle flagged nothing")
                nerr += 1
            if verbose:
                print(f"  (reference raised {type(raised).__name__}: {raised}; status={st}) {tag}")
            sync_reference_from_oracle(ref, orc)
            continue
        if st != 0:
            print(f"MISMATCH {tag}: oracle flagged status {st} but reference did not raise")
            nerr += 1
            sync_reference_from_oracle(ref, orc)
            continue
        ok = compare(ref, orc, tag)
        if int(r_ref) != int(r[0]) or bool(term_ref) != bool(term[0]) or trunc_ref is not False:
            print(f"MISMATCH {tag}: reward {r_ref} vs {r[0]}, terminated {term_ref} vs {term[0]}")
            ok = False
        if info["steps"] != orc.cnt[0, 0] or info.get("submit_count", orc.cnt[0, 1]) != orc.cnt[0, 1]:
            if not variant == "raw" or info["steps"] != orc.cnt[0, 0]:
                print(f"MISMATCH {tag}: steps {info['steps']} vs {orc.cnt[0,0]}, submit_count "
                      f"{info.get('submit_count')} vs {orc.cnt[0,1]}")
                ok = False
        if not ok:
            nerr += 1
            if nerr > 3:
                break
            sync_reference_from_oracle(ref, orc)
    return nerr, nexc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--traces", type=int, default=400)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--big", action="store_true")
    a = ap.parse_args()
    RD.import_reference()
    total = exc = 0
    for tid in range(a.traces):
        e, x = run_trace(tid, a.seed, a.steps, a.verbose, a.big)
        total += e
        exc += x
    print(f"{a.traces} traces x {a.steps} steps: {total} mismatching steps, {exc} steps where the reference raised")
    return 1 if total else 0


if __name__ == "__main__":
    if "--big" in sys.argv:
        import threading
        sys.setrecursionlimit(200000)
        threading.stack_size(1 << 30)
        rc = []
        t = threading.Thread(target=lambda: rc.append(main()))
        t.start()
        t.join()
        sys.exit(rc[0] if rc else 2)
    sys.exit(main())
