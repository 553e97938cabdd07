#!/usr/bin/env python3
"""What would longest-op-first dispatch buy?  The C3 action stream as is, and with each step's ops re-dealt to the envs so that the
object ops (Rotate / Flip, then Move) sit in the workgroups every XCD dispatches FIRST (same multiset of ops per step, same bboxes):
graph-replayed us per launch of 8192 envs.  (A scheduling experiment: a real ordering needs the ops classified before the launch.)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from arcle_amd import actions
from arcle_amd.engine import EnvBatch
from arcle_amd.envs import O2ARCv2Env
dev = torch.device("cuda:0"); n = 8192; K = 400
bbox_np, op_np = bench.make_actions(K, n, 5)
nb = n // 8
vb = np.arange(n) // 8
blk = (vb % (nb // 8)) * 8 + vb // (nb // 8)   # workgroup index of the env's wave (inverse of the kernel's XCD-contiguous map)
pos = blk // 8                                 # position in its XCD's dispatch sequence
cost = np.zeros(64, np.int32); cost[20:24] = 2; cost[24:28] = 3   # Move, Rotate / Flip
xcd = blk % 8
def swapped(tl, te):
    # per XCD: object ops in workgroups at position >= tl trade places with non-object ops at position < te (latest <-> earliest)
    out = op_np.copy()
    for s in range(K):
        o = out[s]
        for x in range(8):
            m = np.nonzero(xcd == x)[0]
            m = m[np.argsort(pos[m], kind="stable")]
            late = m[(pos[m] >= tl) & (cost[o[m]] > 0)][::-1]
            early = m[(pos[m] < te) & (cost[o[m]] == 0)]
            k = min(len(late), len(early))
            a, b = late[:k], early[:k]
            o[a], o[b] = o[b].copy(), o[a].copy()
    return out
def xcd_weighted(weights):
    # object ops first inside every XCD, and dealt to the XCDs in proportion to `weights` (the XCDs start up to 0.7 us apart)
    out = np.empty_like(op_np)
    w = np.asarray(weights, float); w = w / w.sum()
    for s in range(K):
        o = op_np[s]
        lg = o[cost[o] > 0]; sh = o[cost[o] == 0]
        lg = lg[np.argsort(-cost[lg], kind="stable")]
        quota = np.floor(w * len(lg)).astype(int); quota[0] += len(lg) - quota.sum()
        li = si = 0
        for x in range(8):
            m = np.nonzero(xcd == x)[0]
            m = m[np.argsort(pos[m], kind="stable")]
            q = quota[x]
            out[s, m[:q]] = lg[li:li + q]; li += q
            out[s, m[q:]] = sh[si:si + len(m) - q]; si += len(m) - q
    return out
cost3 = cost.copy(); cost3[0:10] = 1   # Color waves live ~0.3 us longer than FloodFill / Copy / Paste / critical ones
def sorted_by(c):
    out = np.empty_like(op_np)
    slots = np.argsort(pos, kind="stable")
    for s in range(K):
        o = op_np[s]
        out[s, slots] = o[np.argsort(-c[o], kind="stable")]
    return out
def variant(v):
    if v == 0: return op_np
    if v == 11: return sorted_by(cost3)
    if v >= 8: return xcd_weighted([(1, 1, 1, 1, 0, 0, 1, 1), (1.2, 1.1, 1, 1, 0.6, 0.6, 0.9, 0.9), (1.5, 1.3, 1.1, 1, 0.3, 0.3, 0.8, 0.8)][v - 8])
    if v >= 3: return swapped(*[(64, 64), (48, 48), (32, 32), (64, 32), (96, 64)][v - 3])
    out = np.empty_like(op_np)
    slots = np.argsort(pos if v == 1 else -pos, kind="stable")     # v=1: long ops first; v=2: long ops LAST (the worst case)
    for s in range(K):
        o = op_np[s]
        out[s, slots] = o[np.argsort(-cost[o], kind="stable")]
    return out
for rep in range(1):
    for v, name in [(i, nm) for i, nm in enumerate(["C3 mix as generated", "object ops dispatched first", "object ops dispatched last", "swap late>=64 early<64", "swap late>=48 early<48", "swap late>=32 early<32", "swap late>=64 early<32", "swap late>=96 early<64", "first + none on XCD 4,5", "first + XCD weights mild", "first + XCD weights strong", "object ops, then Color, then the rest"]) if i in (0, 1, 11)]:
        batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
        batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
        batch.set_tasks_padded(*bench.make_tasks(n, 1)); batch.reset()
        FL = batch.elide_flag | bench.STEP_AUTORESET
        bbox = torch.from_numpy(bbox_np).to(dev); ops = torch.from_numpy(variant(v)).to(dev)
        st = torch.cuda.Stream(dev); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(K):
                batch.step_bbox_ptr(bbox[i].data_ptr(), ops[i].data_ptr(), FL, torch.cuda.current_stream(dev).cuda_stream)
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / K * 1e3)
        print(f"{name:30s} {sorted(ts)[3]:.2f} us per launch", flush=True)
