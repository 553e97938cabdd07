#!/usr/bin/env python3
"""Golden vectors for grids BEYOND 1024 cells, captured from the UNMODIFIED reference (build container only: needs /root/reference).

    python tests/golden/make_golden_big.py

Same capture and file layout as make_golden.py (whose `capture` it calls); the files are named big_*.npz.  They pin the oracle — and
through it the workgroup-per-env kernels of arcle_amd/csrc/arcle_big.h — at max_grid_size values the one-wavefront kernels do not
serve (H * W > 1024; the reference takes any max_grid_size, base.py:37-49).  The reference's flood fill recurses once per cell
(color.py:16-28): the capture raises Python's recursion limit and runs on a thread with a large stack — harness settings, the
reference's code is untouched.
"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as MG  # noqa: E402


def main():
    MG.RD.import_reference()
    MG.capture("big_o2arc_40", "o2arc", 40, 40, -1, 6, 96, 201, full_every=32)
    MG.capture("big_o2arc_64", "o2arc", 64, 64, 3, 3, 64, 202, full_every=32)
    MG.capture("big_o2arc_exotic_45", "o2arc_exotic", 45, 45, -1, 4, 96, 203, full_every=48)
    MG.capture("big_o2arc_crop_50", "o2arc_crop", 50, 50, -1, 4, 64, 204, full_every=32)
    # (non-square planes: the reference's Rotate raises on them, object.py:45 — SURVEY.md A.6-2 — so the ARCEnv table, which has no object ops)
    MG.capture("big_arc_33x100", "arc", 33, 100, 3, 4, 64, 205, full_every=32)
    MG.capture("big_raw_127", "raw", 127, 127, 2, 2, 32, 206, full_every=16)
    MG.capture("big_quirks_40", "o2arc", 40, 40, -1, 4, 300, 207, script=MG.QuirkScript(), full_every=100)
    MG.capture("big_flood_48", "o2arc", 48, 48, -1, 6, 32, 208, script=MG.FloodScript(), full_every=16)


if __name__ == "__main__":
    sys.setrecursionlimit(200000)
    threading.stack_size(1 << 30)
    t = threading.Thread(target=main)
    t.start()
    t.join()
