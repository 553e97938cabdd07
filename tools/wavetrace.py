#!/usr/bin/env python3
"""Diagnostic (needs a -DARCLE_TRACE_WAVES build, ARCLE_HIP_LIB pointing at it): per-wave timestamps (100 MHz realtime
clock) of ONE launch of the C3 mix -> dispatch ramp, per-phase and per-op lifetimes, tail.
  tr[0] kernel entry  tr[1] expansion table built + barrier passed  tr[2] scalar inputs landed
  tr[3] op applied (stores issued)  tr[4] epilogue stores issued"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions, _lib
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; K = 40
b = EnvBatch(n, 30, 30, -1, "o2arc", dev)
b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
b.set_tasks_padded(*bench.make_tasks(n, 1)); b.reset()
bn, on = bench.make_actions(K, n, 5)
bb, oo = torch.from_numpy(bn).to(dev), torch.from_numpy(on).to(dev)
b.enable_accounting(True)
L = _lib.lib()
L.arcle_debug_copy_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
cls = {"color": range(0, 10), "floodfill": range(10, 20), "move": range(20, 24), "rot/flip": range(24, 28),
       "copy/paste": range(28, 31), "critical": range(31, 34), "submit": [34]}
for rep in range(3):
    for i in range(K - 1):
        b.step_bbox(bb[i], oo[i], b.elide_flag)
    torch.cuda.synchronize()
    b.step_bbox(bb[K - 1], oo[K - 1], b.elide_flag)
    tr = np.zeros((n, 8), np.uint64)
    assert L.arcle_debug_copy_trace(b._h, tr.ctypes.data) == 0
    t0 = tr[:, 0].min()
    T = (tr[:, :5] - t0).astype(np.float64) / 100.0  # us
    st, en = T[:, 0], T[:, 4]
    ph = np.diff(T, axis=1)  # lut, inputs, core, epilogue
    life = en - st
    ops = on[K - 1]
    print(f"rep {rep}: last start {st.max():.2f} us, last end {en.max():.2f} us; lifetime mean {life.mean():.2f} "
          f"p50 {np.median(life):.2f} p90 {np.percentile(life,90):.2f} max {life.max():.2f}")
    nb = n // 8
    vb = np.arange(n) // 8
    blk = (vb % (nb // 8)) * 8 + vb // (nb // 8)  # inverse of vb = (b & 7) * (nb / 8) + (b >> 3): the workgroup index an env's wave ran in
    order = np.argsort(blk, kind="stable")
    chunks = np.array_split(st[order], 8)
    print("   start by workgroup-index octile (us, mean): " + " ".join(f"{c.mean():.2f}" for c in chunks) + f"   corr(start, wg index) = {np.corrcoef(st, blk)[0, 1]:.3f}")
    xcd, pos = blk % 8, blk // 8
    print("   start mean by XCD (wg index % 8): " + " ".join(f"{st[xcd == x].mean():.2f}" for x in range(8)) +
          "   corr(start, position in the XCD's sequence) per XCD: " + " ".join(f"{np.corrcoef(st[xcd == x], pos[xcd == x])[0, 1]:.2f}" for x in range(8)))
    x0 = xcd == 0
    ch = np.array_split(st[x0][np.argsort(pos[x0], kind="stable")], 16)
    print("   XCD 0, start by position 1/16ths (mean / min / max): " + " ".join(f"{c.mean():.2f}/{c.min():.2f}/{c.max():.2f}" for c in ch))
    print("   start percentiles (us): " + " ".join(f"p{q}={np.percentile(st,q):.2f}" for q in (10, 50, 90, 99)))
    print("   end   percentiles (us): " + " ".join(f"p{q}={np.percentile(en,q):.2f}" for q in (10, 50, 90, 99)))
    print("   phases mean (us): table+barrier %.2f | inputs landed %.2f | op %.2f | epilogue %.2f" % tuple(ph.mean(0)))
    for k, r in cls.items():
        m = np.isin(ops, list(r))
        print(f"   {k:10s} n={m.sum():5d} life mean {life[m].mean():5.2f} p90 {np.percentile(life[m],90):5.2f}  end mean {en[m].mean():5.2f} max {en[m].max():5.2f}"
              f" | lut {ph[m,0].mean():4.2f} in {ph[m,1].mean():4.2f} op {ph[m,2].mean():4.2f} epi {ph[m,3].mean():4.2f}")
