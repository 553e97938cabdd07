#!/usr/bin/env python3
"""Long differential soak on the GPU (not part of the test suite): random tasks / actions / ingress forms, every state field
compared with the oracle after every step — the lean 30x30 instantiations (runtime flags, compile-time flags, fused packed rows),
a non-30x30 FW_FULL/FAST shape and a generic-width shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import backends as B
from oracle import oracle as O
ops = O.o2arc_ops()
S = int(os.environ.get("SOAK_STEPS", 600))
BE = getattr(B, os.environ.get("SOAK_BACKEND", "HipBackend"))  # (EmuBackend: a dry run of this script on the CPU)
SCALE = int(os.environ.get("SOAK_SHRINK", 1))
t0 = time.time()
for name, H, W, N, flags, mt in (("30x30 flags=0", 30, 30, 1024, 0, 3), ("30x30 autoreset|elide", 30, 30, 1024, 1 | 2, 3),
                                 ("30x30 autoreset|elide, max_trial=-1", 30, 30, 1024, 1 | 2, -1),
                                 ("24x32 autoreset", 24, 32, 512, 1, 2), ("17x21 autoreset|elide", 17, 21, 512, 1 | 2, 5),
                                 ("9x13 generic", 9, 13, 512, 1, 3)):
    errs = B.random_trace_compare(BE, "o2arc", ops, H, W, N=N // SCALE, S=S, seed=H * 131 + W + flags, max_trial=mt, flags=flags,
                                  bad_ops=True)
    print(f"{name:40s} N={N} S={S}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
# round 5: grids beyond 1024 cells (one workgroup per env, arcle_big.hip; SOAK_BACKEND=BigEmuBackend is the CPU dry run of these blocks)
if BE is not getattr(B, "EmuBackend"):
    O.set_threads(8)
    for name, H, W, N, flags, mt in (("40x40 big-grid autoreset|elide", 40, 40, 256, 1 | 2, 3), ("64x64 big-grid flags=0", 64, 64, 128, 0, -1),
                                     ("127x127 big-grid autoreset", 127, 127, 64, 1, 3), ("33x100 big-grid autoreset|elide", 33, 100, 128, 1 | 2, 2)):
        errs = B.random_trace_compare(BE, "o2arc", ops, H, W, N=max(N // SCALE, 2), S=max(S // 4, 8), seed=H * 131 + W + flags, max_trial=mt, flags=flags,
                                      bad_ops=True)
        print(f"{name:40s} N={N} S={max(S // 4, 8)}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
    # round 6: the mask ingest on whole words — masks of arbitrary int8 values, and the record / bit-row forms — through the two-chunk kernels
    for name, H, W, N, kw in (("40x40 big-grid int8 masks of any value", 40, 40, 256, dict(int8_masks=True)), ("64x48 big-grid int8 masks of any value", 64, 48, 128, dict(int8_masks=True)),
                              ("127x127 big-grid int8 masks of any value", 127, 127, 32, dict(int8_masks=True)), ("50x50 big-grid bbox5 / bit rows", 50, 50, 128, dict(new_forms=True))):
        errs = B.random_trace_compare(BE, "o2arc", ops, H, W, N=max(N // SCALE, 2), S=max(S // 4, 8), seed=H * 7 + W, max_trial=3, flags=3, **kw)
        print(f"{name:40s} N={N} S={max(S // 4, 8)}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
    O.set_threads(1)
if os.environ.get("SOAK_BACKEND") == "BigEmuBackend":
    sys.exit(0)
# round 4: batches of at most 2048 envs take the small-batch paths (speculative grid request, 256-thread workgroups) — the blocks above.
# The plain lean kernel that the headline times (4096 envs: beyond that threshold) and every speculative policy forced onto the same size:
O.set_threads(8)
for pol in (None, "A", "B", "H", "J"):
    if pol is None:
        os.environ.pop("ARCLE_STREAM_POLICY", None)
    else:
        os.environ["ARCLE_STREAM_POLICY"] = pol
    errs = B.random_trace_compare(BE, "o2arc", ops, 30, 30, N=4096 // SCALE, S=max(S // 6, 8), seed=4242, max_trial=3, flags=3, bad_ops=True)
    print(f"{'30x30 autoreset|elide, policy ' + str(pol):40s} N=4096 S={max(S // 6, 8)}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
os.environ.pop("ARCLE_STREAM_POLICY", None)
O.set_threads(1)
arc = O.arc_ops()
errs = B.random_trace_compare(BE, "arc", arc, 30, 30, N=1024 // SCALE, S=S, seed=77, max_trial=3, flags=1, op_weights=[1] * 10 + [7] * 10 + [1] * 7)
print(f"{'ARCEnv 30x30 FloodFill-heavy':40s} N=1024 S={S}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)

# round-3 forms: 5-tuple records + bit-packed masks through the lean instantiations, and the research flag set (dense pair cache,
# fused FilterO2ARC rows kept incrementally, auto-reset, elision) checked against the oracle's state and the stand-alone row writer
import numpy as np
import rows as R
for name, flags in (("30x30 bbox5/bits flags=0", 0), ("30x30 bbox5/bits autoreset|elide", 3)):
    errs = B.random_trace_compare(BE, "o2arc", ops, 30, 30, N=1024 // SCALE, S=S, seed=911 + flags, max_trial=3, flags=flags, new_forms=True)
    print(f"{name:40s} N=1024 S={S}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
N, H, W = 512 // SCALE, 30, 30
FL = R.STEP_AUTORESET | R.STEP_ELIDE | R.STEP_DENSE | R.STEP_FLAT_OBS
for filtered in (True, False):
    rng = np.random.default_rng(5 + filtered)
    be, orc = BE(N, H, W, 3, "o2arc", ops), B.OracleBackend(N, H, W, 3, "o2arc", ops)
    tasks = R._tasks(rng, N, H, W, same_answer=0.7)
    for b in (be, orc):
        b.set_tasks(*tasks)
        b.reset()
    be.set_dense_output()
    be.set_flat_output(filtered)
    be.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 32, np.int32), FL)
    orc.step("bbox", np.zeros((N, 4), np.int32), np.full(N, 32, np.int32), R.STEP_AUTORESET)
    errs = []
    fields = R._state_fields("o2arc")
    for s in range(S):
        ing, pay, op = R._random_actions(rng, N, H, W, len(ops))
        r1, t1 = be.step(ing, pay, op, FL | R.STEP_ROWS_INC)
        r2, t2 = orc.step(ing, pay, op, R.STEP_AUTORESET)
        if not (np.array_equal(r1, r2) and np.array_equal(t1, t2)):
            errs.append(f"step {s}: reward / terminated differ")
        for f in fields:
            if not np.array_equal(be.get(f), orc.get(f)):
                errs.append(f"step {s}: field {f} differs")
        if not np.array_equal(be.fused_flat(), be.flat_obs(filtered)):
            errs.append(f"step {s}: incremental rows differ from the stand-alone writer's")
        if s % 8 == 0:
            gd, ad = orc.get("grid_dim").astype(int), orc.get("answer_dim").astype(int)
            g, a, d = orc.get("grid"), orc.get("answer"), be.dense
            for n in range(0, N, 7):
                mh, mw = min(gd[n, 0], ad[n, 0]), min(gd[n, 1], ad[n, 1])
                correct = int((g[n, :mh, :mw] == a[n, :mh, :mw]).sum())
                if int(d[n, 0]) != correct and tuple(d[n]) != (0, 0):
                    errs.append(f"step {s} env {n}: dense correct {d[n, 0]} != {correct}")
        be.status(), orc.status()
        if errs:
            break
    print(f"{'research flags, filtered=' + str(filtered):40s} N={N} S={S}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
