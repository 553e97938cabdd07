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
t0 = time.time()
for name, H, W, N, flags, mt in (("30x30 flags=0", 30, 30, 1024, 0, 3), ("30x30 autoreset|elide", 30, 30, 1024, 1 | 2, 3),
                                 ("30x30 autoreset|elide, max_trial=-1", 30, 30, 1024, 1 | 2, -1),
                                 ("24x32 autoreset", 24, 32, 512, 1, 2), ("17x21 autoreset|elide", 17, 21, 512, 1 | 2, 5),
                                 ("9x13 generic", 9, 13, 512, 1, 3)):
    errs = B.random_trace_compare(B.HipBackend, "o2arc", ops, H, W, N=N, S=S, seed=H * 131 + W + flags, max_trial=mt, flags=flags,
                                  bad_ops=True)
    print(f"{name:40s} N={N} S={S}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
arc = O.arc_ops()
errs = B.random_trace_compare(B.HipBackend, "arc", arc, 30, 30, N=1024, S=S, seed=77, max_trial=3, flags=1, op_weights=[1] * 10 + [7] * 10 + [1] * 7)
print(f"{'ARCEnv 30x30 FloodFill-heavy':40s} N=1024 S={S}: {'OK' if not errs else errs[:3]}  ({time.time() - t0:.0f} s)", flush=True)
