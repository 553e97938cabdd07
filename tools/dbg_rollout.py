import sys, os, subprocess
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
if len(sys.argv) > 1:
    import numpy as np
    import backends as B
    from oracle import oracle as O
    H, W, ing, N, T, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    if kind == "exotic":
        from oracle import refdriver as RD
        ops = RD.variant_table("o2arc_exotic")[1]; kind = "o2arc"
    elif kind == "o2sub":   # o2arc planes, a non-canonical but plain table (first 27 ops)
        ops = O.o2arc_ops()[:34] + [O.desc(O.OP_SUBMIT)]; ops[33] = O.desc(O.OP_CROP_GRID, 0, 1); kind = "o2arc"
    else:
        ops = O.KIND_OPS[kind]()
    errs = B.rollout_compare(B.HipBackend, kind, ops, H, W, N=N, T=T, seed=1, ingress=ing)
    print(" ->", errs[:3], flush=True)
else:
    for cfg in ["10 10 bbox 64 16 exotic", "10 10 bbox 64 16 o2sub", "10 10 bbox 64 1 arc", "10 10 bbox 1 2 arc", "10 10 bbox 64 2 raw", "10 10 point 64 16 raw", "10 10 bbox 64 16 o2arc"]:
        r = subprocess.run([sys.executable, __file__] + cfg.split(), capture_output=True, text=True)
        print(cfg, "rc", r.returncode, (r.stdout + r.stderr).strip().splitlines()[-1][:200], flush=True)
