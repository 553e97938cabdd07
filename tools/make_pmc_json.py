#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc.sh, csv output) into
profiles/pmc_latest.json: HBM bytes per launch of arcle_step_kernel.  Corrections as prescribed by
/opt/skills/guides/MI355X_MICROARCH.md §HBM: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide coalesced (16 B/lane) read stream, so it is doubled; WRITE_SIZE is taken as is."""
import csv, glob, json, os, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
tag = sys.argv[2] if len(sys.argv) > 2 else "round2"


def mean_counter(d, name):
    vals = []
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "arcle_step" in row.get("Kernel_Name", "") and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch, n1 = mean_counter("pmc_mem1", "FETCH_SIZE")
write, n2 = mean_counter("pmc_mem2", "WRITE_SIZE")
out = {"kernel": "arcle_step_kernel", "launches_sampled": [n1, n2], "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "read_bytes_per_launch": fetch * 1024 * 2, "write_bytes_per_launch": write * 1024,
       "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
       "source": f"profiles/{tag}_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                 "`python bench.py --steps 40 --warmup 5 --no-graph --no-ramp --regions 2`; FETCH_SIZE x2 (gfx950 half-count "
                 "of 16 B/lane streams), KiB->B"}
json.dump(out, open("profiles/pmc_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1))
