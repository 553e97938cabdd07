#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd .db (kernel trace): per-kernel call count / avg / median / min / max duration,
optionally in consecutive chunks of the arcle_step_kernel launches (--chunk N) for staged experiments, and
(--timed K R) the launches of bench.py's timed regions alone: its R hipGraph replays of K steps are the LAST K*R launches of
the most-launched step-kernel instantiation — everything before them is the untimed clock ramp / warm-up / graph upload."""
import sqlite3
import sys

import numpy as np

db = sys.argv[1]
chunk = int(sys.argv[sys.argv.index("--chunk") + 1]) if "--chunk" in sys.argv else 0
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
by = {}
for name, s, e in rows:
    by.setdefault(name.split("(")[0], []).append((e - s) / 1e3)
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>8s} {'med_us':>8s} {'min_us':>8s} {'max_us':>8s} {'total_ms':>9s}")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print(f"{k[-60:]:60s} {len(v):6d} {v.mean():8.2f} {np.median(v):8.2f} {v.min():8.2f} {v.max():8.2f} {v.sum() / 1e3:9.3f}")
if chunk:
    v = np.array([d for name, s, e in rows if "arcle_step_kernel" in name for d in [(e - s) / 1e3]])
    for i in range(0, len(v), chunk):
        w = v[i:i + chunk]
        print(f"step launches {i:5d}..{i + len(w) - 1:5d}: avg {w.mean():7.2f} us  median {np.median(w):7.2f}  min {w.min():7.2f}")
if "--timed" in sys.argv:
    K, R = int(sys.argv[sys.argv.index("--timed") + 1]), int(sys.argv[sys.argv.index("--timed") + 2])
    # the benchmark's instantiation = the step-kernel instantiation with the most launches (the byte-accounting one runs once per action batch)
    counts = {}
    for name, s, e in rows:
        if "arcle_step_kernel" in name:
            counts[name.split("(")[0]] = counts.get(name.split("(")[0], 0) + 1
    hot = max(counts, key=counts.get)
    plain = [(name, (e - s) / 1e3) for name, s, e in rows if name.split("(")[0] == hot]
    v = np.array([d for _, d in plain][-K * R:])
    print(f"timed regions ({R} x {K} launches of {plain[-1][0].split('(')[0]}): avg {v.mean():.2f} us  median {np.median(v):.2f}  "
          f"min {v.min():.2f}  max {v.max():.2f}")
