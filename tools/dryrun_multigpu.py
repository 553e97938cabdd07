#!/usr/bin/env python3
"""Dry run of tests/test_multi_gpu_hip.py on a ONE-GPU box: the test's own worker with world size 1 over RCCL (sync / overlapped / ping-pong /
every-4 / graph-captured gathers) against the oracle — shakes out the test itself before it ever meets a multi-GPU node."""
import sys, os, queue
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_multi_gpu_hip as T
import backends as B
from oracle import oracle as O
class Q:
    def put(self, x): self.x = x
q = Q()
G = 1 * 2304 + 3
T._worker(0, 1, T._free_port(), G, q)
out = q.x
tasks, bb_np, op_np = T._inputs(G)
O.set_threads(16)
orc = B.OracleBackend(G, 30, 30, -1, "o2arc", O.o2arc_ops()); orc.set_tasks(*tasks); orc.reset()
want, rew = [], []
for s in range(T.S):
    r, t = orc.step("bbox", bb_np[s], op_np[s], O.STEP_AUTORESET)
    want.append(orc.get("grid").copy()); rew.append(r.copy())
want, rew = np.stack(want), np.stack(rew)
print("sync", np.array_equal(out["sync"][0], want), np.array_equal(out["sync"][1], rew))
print("async", np.array_equal(out["async"], want))
(ga, gb), (ia, ib) = out["groups"]
print("groups", np.array_equal(ga, want[:, ia]) and np.array_equal(gb, want[:, ib]), sorted(ia.tolist()+ib.tolist()) == list(range(G)))
print("every", np.array_equal(out["every"], want))
print("captured", out["captured_mode"], None if out["captured"] is None else np.array_equal(out["captured"], want[3]))
print("ranks", out["ranks_seen"])
