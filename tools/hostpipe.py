#!/usr/bin/env python3
"""Host-resident 20-byte action records: the copy of step t+1's records (pinned host -> device, double-buffered) on a side stream
overlapping the step kernel of step t, all inside one hipGraph — against the zero-copy form (the kernel reads pinned host memory)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0"); n = 8192; K = 200
bbox_np, op_np = bench.make_actions(K, n, 2000)
act5 = torch.from_numpy(np.concatenate([bbox_np, op_np[..., None]], -1).astype(np.int32))
h5 = act5.pin_memory()
batch = bench.make_batch(dev, n)
FL = batch.elide_flag | bench.STEP_AUTORESET
L, h = batch.L, batch._h
d5 = [torch.empty((n, 5), dtype=torch.int32, device=dev) for _ in range(2)]


def pipelined(sh):
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    copied, stepped = [None, None], [None, None]
    side.wait_stream(main)
    for i in range(K):
        s = i & 1
        with torch.cuda.stream(side):
            if stepped[s] is not None:
                side.wait_event(stepped[s])  # the kernel that read this staging buffer two steps ago
            d5[s].copy_(h5[i], non_blocking=True)
            copied[s] = torch.cuda.Event(); copied[s].record(side)
        main.wait_event(copied[s])
        assert L.arcle_step_bbox5(h, d5[s].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, main.cuda_stream) == 0
        stepped[s] = torch.cuda.Event(); stepped[s].record(main)
    main.wait_stream(side)


def zero_copy(sh):
    for i in range(K):
        assert L.arcle_step_bbox5(h, h5[i].data_ptr(), batch._reward_ptr, batch._term_ptr, FL, sh) == 0


for name, fn in (("zero-copy", zero_copy), ("copy pipelined on a side stream", pipelined), ("zero-copy", zero_copy)):
    sec, _ = bench.graph_time(dev, fn, K)
    print(f"{name:36s} {sec * 1e6:6.2f} us per step", flush=True)
