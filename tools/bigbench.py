#!/usr/bin/env python3
"""Grids beyond 1024 cells (the workgroup-per-env kernels, arcle_amd/csrc/arcle_big.hip): us per step of a batch, graph-replayed, HIP events.
    python tools/bigbench.py [--sizes 40x40,64x64,127x127] [--envs 1024,4096] [--steps 24]
The C3 action mix (35 ops uniform, BBox tuples uniform over the plane) on O2ARCv2Env tasks of the given max_grid_size; next to the time, the
plane traffic a step of that op would move by the per-op model of bench.py's big_grid leg."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="40x40,64x64,127x127")
    ap.add_argument("--envs", default="1024,4096")
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--ops", default="", help="restrict the op indices drawn, e.g. 10-19 (FloodFill) or 20-23 (Move); default: all 35")
    ap.add_argument("--ingress", default="bbox", choices=["bbox", "mask", "bits"], help="the same rectangles as tuples / full int8 masks / bit-packed masks")
    ap.add_argument("--point-seeds", action="store_true", help="every rectangle a single cell (FloodFill really fills)")
    ap.add_argument("--eager", action="store_true", help="plain launches instead of graph replays (PMC passes: rocprofv3 counts no graph-launched kernels)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.eager:
        for size in a.sizes.split(","):
            H, W = (int(v) for v in size.split("x"))
            for n in (int(v) for v in a.envs.split(",")):
                batch = BN.make_batch(dev, n, 1000, "o2arc", H, W)
                bb, oo = BN.make_actions(a.steps, n, 2000, H, W)
                if a.ops:
                    lo, hi = (int(v) for v in a.ops.split("-"))
                    oo = (lo + oo % (hi - lo + 1)).astype(np.int32)
                if a.point_seeds:
                    bb[..., 2:] = bb[..., :2]
                bbd, ood = torch.from_numpy(bb).to(dev), torch.from_numpy(oo).to(dev)
                if a.ingress != "bbox":  # the same rectangles as int8 masks / bit rows
                    x1, x2 = torch.minimum(bbd[..., 0], bbd[..., 2]), torch.maximum(bbd[..., 0], bbd[..., 2])
                    y1, y2 = torch.minimum(bbd[..., 1], bbd[..., 3]), torch.maximum(bbd[..., 1], bbd[..., 3])
                    ii, jj = torch.arange(H, device=dev)[None, :, None], torch.arange(W, device=dev)[None, None, :]
                for i in range(a.steps):
                    if a.ingress == "bbox":
                        batch.step_bbox_ptr(bbd[i].data_ptr(), ood[i].data_ptr(), batch.elide_flag | 1, torch.cuda.current_stream(dev).cuda_stream)
                    else:
                        m = ((ii >= x1[i, :, None, None]) & (ii <= x2[i, :, None, None]) & (jj >= y1[i, :, None, None]) & (jj <= y2[i, :, None, None])).to(torch.int8).contiguous()
                        if a.ingress == "bits":
                            batch.step_bits(batch.pack_mask_bits(m), ood[i], batch.elide_flag | 1)
                        else:
                            batch.step_mask(m, ood[i], batch.elide_flag | 1)
                torch.cuda.synchronize()
                print(f"{H}x{W} envs {n}: {a.steps} eager steps", flush=True)
        return
    for size in a.sizes.split(","):
        H, W = (int(v) for v in size.split("x"))
        for n in (int(v) for v in a.envs.split(",")):
            ops = None
            if a.ops:
                lo, hi = (int(v) for v in a.ops.split("-"))
                ops = (lo, hi)
            leg = BN.big_grid_case(dev, H, W, n, a.steps, ops=ops, ingress=a.ingress, point_seeds=a.point_seeds)
            rl = leg["roofline"]
            print(f"{H}x{W} envs {n}{' ops ' + a.ops if a.ops else ''}{' ' + a.ingress if a.ingress != 'bbox' else ''}: {leg['us_per_step_batch']:.1f} us per step = {leg['value'] / 1e6:.1f} M env-steps/s; kernel-counted "
                  f"{rl['algorithmic_bytes_per_launch'] / 1e6:.1f} MB per launch ({rl['traffic'] / 1e6:.1f} issued) -> {rl['frac']:.3f} of 8 TB/s "
                  f"({rl['frac_by_traffic']:.3f} by issued bytes; modelled {rl['modelled_bytes_per_launch'] / 1e6:.1f} MB)", flush=True)


if __name__ == "__main__":
    main()
