#!/usr/bin/env python3
"""arcle_step_many with and without the ordered dispatch (object operations handed to the waves that start first): identical results,
us per step of 8192 envs when the K steps are replayed as one hipGraph."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192; K = 200
bbox_np, op_np = bench.make_actions(K, n, 5)
bbox, ops = torch.from_numpy(bbox_np).to(dev), torch.from_numpy(op_np).to(dev)
rec5 = torch.cat([bbox, ops[:, :, None]], 2).contiguous()
def make(ordered):
    b = EnvBatch(n, 30, 30, -1, "o2arc", dev)
    b.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
    b.set_tasks_padded(*bench.make_tasks(n, 1)); b.reset()
    b.set_dispatch_order(ordered)
    return b
# the table launch 0 wrote for step 1: a permutation inside every XCD's range with the object ops in the lowest slots
import ctypes
b = make(True)
b.step_many("bbox", bbox[:2].contiguous(), ops[:2].contiguous(), b.elide_flag | bench.STEP_AUTORESET)
tab = np.zeros((3, n), np.uint32)
b.L.arcle_debug_copy_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert b.L.arcle_debug_copy_order(b._h, tab.ctypes.data) == 0
t1, rs = tab[1].astype(np.int64), n // 8
lg = (op_np[1] >= 20) & (op_np[1] < 28)
for x in range(8):
    seg = t1[x * rs:(x + 1) * rs]
    assert sorted(seg.tolist()) == list(range(x * rs, (x + 1) * rs)), f"XCD {x}: not a permutation of its range"
    L = int(lg[x * rs:(x + 1) * rs].sum())
    assert lg[seg[:L]].all() and not lg[seg[L:]].any(), f"XCD {x}: object ops are not in the first {L} slots"
print("order table of step 1: a permutation per XCD range, object ops in the first slots; moved slots:", int((t1 != np.arange(n)).sum()), "identity table ok:", bool((tab[2] == np.arange(n)).all()), flush=True)
pre = op_np.copy()   # the same ops dealt so that no slot has to move
for s in range(K):
    for x in range(8):
        o = pre[s, x * rs:(x + 1) * rs]
        pre[s, x * rs:(x + 1) * rs] = o[np.argsort(~((o >= 20) & (o < 28)), kind="stable")]
ops_pre = torch.from_numpy(pre).to(dev)
res = {}
for form in ("bbox", "bbox-presorted") + (("bbox5",) if not os.environ.get("ORD_FAST") else ()):
    for ordered in (False, True):
        b = make(ordered)
        FL = b.elide_flag | bench.STEP_AUTORESET
        pay, op = (bbox, ops) if form == "bbox" else (bbox, ops_pre) if form == "bbox-presorted" else (rec5, None)
        form_ = "bbox5" if form == "bbox5" else "bbox"
        r, t = b.step_many(form_, pay, op, FL)
        torch.cuda.synchronize()
        state = [b.planes[k].cpu().numpy().copy() for k in sorted(b.planes)] if hasattr(b, "planes") else []
        res[(form, ordered)] = (r.cpu().numpy(), t.cpu().numpy(), b.get_state_rows().cpu().numpy(), b.status())
        st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
        rr, tt = torch.empty_like(r), torch.empty_like(t)
        with torch.cuda.graph(g, stream=st):
            b.step_many(form_, pay, op, FL, rr, tt)
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / K * 1e3)
        print(f"{form:15s} ordered={ordered!s:5s} {sorted(ts)[4]:.2f} us per step", flush=True)
    a, c = res[(form, False)], res[(form, True)]
    same = all(np.array_equal(x, y) for x, y in zip(a[:3], c[:3])) and a[3] == c[3]
    print(f"{form}: ordered == unordered results (reward, terminated, every state row, status): {same}", flush=True)
