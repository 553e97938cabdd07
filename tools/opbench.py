#!/usr/bin/env python3
"""Per-operation kernel time on the GPU: every env executes the SAME op (random bboxes), K launches timed
with one HIP-event pair on the launch stream.  Diagnostic tool for kernel tuning."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from arcle_amd import actions  # noqa: E402
from arcle_amd.engine import EnvBatch  # noqa: E402
from arcle_amd.envs import O2ARCv2Env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--ops", type=str, default="0,10,20,24,26,28,29,30,31,32,33,34,mix")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, K = a.n, a.k
    names = ["".join(map(str.capitalize, o.__name__.split("_"))) for o in O2ARCv2Env.default_operations()]
    bbox_np, op_np = bench.make_actions(K, n, 5)
    bbox = torch.from_numpy(bbox_np).to(dev)
    small = bbox.clone()
    small[..., 2] = torch.minimum(small[..., 0] + 3, torch.tensor(29, device=dev))
    small[..., 3] = torch.minimum(small[..., 1] + 3, torch.tensor(29, device=dev))
    point = bbox.clone()
    point[..., 2:] = point[..., :2]
    empty = torch.full_like(bbox, 40)
    res = {}
    for spec in a.ops.split(","):
        for selname, bb in (("rect", bbox), ("small", small), ("point", point), ("empty", empty)):
            if spec == "mix" and selname != "rect":
                continue
            batch = EnvBatch(n, 30, 30, -1, "o2arc", dev)
            batch.set_op_table(actions.table_descs(O2ARCv2Env.default_operations()))
            batch.set_tasks_padded(*bench.make_tasks(n, 1))
            batch.reset()
            ops = torch.from_numpy(op_np).to(dev) if spec == "mix" else torch.full((K, n), int(spec), dtype=torch.int32, device=dev)
            if selname == "empty" and spec != "mix" and int(spec) in range(20, 28):
                # continued object op: lift an object first
                batch.step_bbox(small[0], torch.full((n,), 20, dtype=torch.int32, device=dev))
            st = torch.cuda.current_stream(dev)
            sh = st.cuda_stream
            for i in range(10):
                batch.step_bbox_ptr(bb[i].data_ptr(), ops[i].data_ptr(), 0, sh)
            batch.enable_accounting(True)
            batch.accounting(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(K):
                batch.step_bbox_ptr(bb[i].data_ptr(), ops[i].data_ptr(), 0, sh)
            e1.record(st)
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / K
            nb, ns = batch.accounting(True)
            key = f"{spec if spec == 'mix' else names[int(spec)]}/{selname}"
            res[key] = {"us_per_launch": round(us, 2), "bytes_per_env_step": round(nb / max(ns, 1), 1),
                        "GBps": round(nb / K / us / 1e3, 1)}
            print(key, res[key], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/opbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
