#!/usr/bin/env python3
"""Diagnostic (needs a -DARCLE_TRACE_WAVES build, ARCLE_HIP_LIB pointing at it): what ONE launch of the headline kernel takes in the regime
bench.py times — K single-step calls replayed back to back as a hipGraph, no tracer attached.  Every wave of the last two launches leaves
its entry / exit time (s_memrealtime, 100 MHz); per launch: first wave in -> last wave out (= the kernel's duration as rocprofv3 would see it if
it did not serialise the dispatches), the gap to the next launch's first wave, and the start-to-start period (= what HIP events / K measure)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import _lib
dev = torch.device("cuda:0"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; K = 40
bn, on = bench.make_actions(K, n, 5)
bb, oo = torch.from_numpy(bn).to(dev), torch.from_numpy(on).to(dev)
L = _lib.lib()
L.arcle_debug_launch_trace.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
for order in (True, False):
    b = bench.make_batch(dev, n, seed=11)
    b.set_dispatch_order(order)
    FL = b.elide_flag | 1
    assert L.arcle_debug_launch_trace(b._h, 1, None, None) == 0
    st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        cs = torch.cuda.current_stream(dev).cuda_stream
        for s in range(K):
            b.step_bbox_ptr(bb[s].data_ptr(), oo[s].data_ptr(), FL, cs)
    res = []
    for rep in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        tr = np.zeros((2, n, 2), np.uint64); seq = ctypes.c_int(0)
        assert L.arcle_debug_launch_trace(b._h, 1, tr.ctypes.data, ctypes.byref(seq)) == 0
        last = seq.value  # (numbers baked in at capture: K-1 is the graph's last launch)
        A, B = tr[(last - 1) & 1].astype(np.float64) / 100.0, tr[last & 1].astype(np.float64) / 100.0  # us
        if rep >= 2:
            res.append((A[:, 1].max() - A[:, 0].min(), B[:, 1].max() - B[:, 0].min(), B[:, 0].min() - A[:, 1].max(), B[:, 0].min() - A[:, 0].min(),
                        e0.elapsed_time(e1) * 1e3 / K))
    r = np.array(res)
    name = "self-ordering <..., autoreset|elide|grouped, 30>" if order else "dispatch order off <..., autoreset|elide, 30>"
    print(f"{name}: first wave in -> last wave out: launch K-2 {np.median(r[:,0]):.2f} us, launch K-1 {np.median(r[:,1]):.2f} us; "
          f"last wave of K-2 out -> first wave of K-1 in: {np.median(r[:,2]):.2f} us; start-to-start {np.median(r[:,3]):.2f} us; "
          f"HIP events / K (same replays): {np.median(r[:,4]):.2f} us", flush=True)
