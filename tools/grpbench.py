#!/usr/bin/env python3
"""The launch that orders itself (ARCLE_STEPX_GROUPED), diagnostics: us per step of 8192 envs (K single-step calls replayed as one hipGraph)
with the natural C3 op stream, with the ops pre-sorted inside every group (no slot trades: the pure overhead of the group logic) and inside
every XCD range (the ideal order for a launch that does NOT order itself); results equal to the ungrouped launches.
  ARCLE_HIP_LIB=... python tools/grpbench.py [n_envs] [group size 16|32|64]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; GS = int(sys.argv[2]) if len(sys.argv) > 2 else 16; K = 200
bbox_np, op_np = bench.make_actions(K, n, 5)
lg = lambda o: (o >= 20) & (o < 28)
rs = n // 8
def presort(kind):
    out = op_np.copy()
    for s in range(K):
        o = out[s]
        if kind == "xcd":
            v = o.reshape(8, rs)
        else:
            v = o.reshape(n // GS, GS)
        idx = np.argsort(~lg(v), axis=1, kind="stable")
        v = np.take_along_axis(v, idx, 1)
        out[s] = v.reshape(n)
    return out
def uniform_l4():  # every group: object ops in its first quarter, others behind — no trades, no group needs more than a quarter of the strata
    out = op_np.copy()
    rng = np.random.default_rng(3)
    for s in range(K):
        v = out[s].reshape(n // GS, GS)
        v[:, :GS // 4] = rng.integers(20, 28, (n // GS, GS // 4))
        short = np.r_[0:20, 28:35]
        v[:, GS // 4:] = short[rng.integers(0, len(short), (n // GS, GS - GS // 4))]
    return out
streams = {"natural": op_np, "group-presorted": presort("grp"), "xcd-presorted": presort("xcd")}
if os.environ.get("GRP_L4"): streams["uniform-L4"] = uniform_l4()
if os.environ.get("GRP_ONLY"): streams = {k: v for k, v in streams.items() if k in os.environ["GRP_ONLY"].split(",")}
bbox = torch.from_numpy(bbox_np).to(dev)
def make(grouped):
    os.environ["ARCLE_GROUPED"] = "1" if grouped else "0"
    b = bench.make_batch(dev, n, seed=11)
    return b
ref = {}
for name, o_np in streams.items():
    ops = torch.from_numpy(o_np).to(dev)
    for grouped in (False, True):
        b = make(grouped)
        FL = b.elide_flag | bench.STEP_AUTORESET
        sh = torch.cuda.current_stream(dev).cuda_stream
        for s in range(K):
            b.step_bbox_ptr(bbox[s].data_ptr(), ops[s].data_ptr(), FL, sh)
        torch.cuda.synchronize()
        rows = b.get_state_rows().cpu().numpy()
        if not grouped: ref[name] = rows
        else: print(f"{name}: grouped == ungrouped state rows: {np.array_equal(rows, ref[name])}  status {b.status()}", flush=True)
        st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            cs = torch.cuda.current_stream(dev).cuda_stream
            for s in range(K):
                b.step_bbox_ptr(bbox[s].data_ptr(), ops[s].data_ptr(), FL, cs)
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(11):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / K * 1e3)
        print(f"  {name:16s} grouped={int(grouped)} {sorted(ts)[5]:.3f} us per step (min {min(ts):.3f})", flush=True)
