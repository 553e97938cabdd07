#!/bin/bash
# per-op instruction counts: PMC pass over tools/opprof.py (100 launches per op, chunked in order)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_op
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --kernel-trace -d $R/gpurun_out/pmc_op -o p --output-format csv -- python $R/tools/opprof.py "$@" 2>&1 | grep chunk | tr '\n' ' '; echo
python - $R/gpurun_out/pmc_op <<'PY'
import sys, csv, glob, collections
d=sys.argv[1]
rows=[]
for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "arcle_step" in row["Kernel_Name"]:
            rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
ids=sorted(set(r[0] for r in rows))
idx={d:i for i,d in enumerate(ids)}
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d_,n,v in rows:
    acc[idx[d_]//100][n].append(v)
names=sorted({r[1] for r in rows})
print("chunk " + " ".join(f"{n.replace('SQ_',''):>14s}" for n in names) + "   (per wave)")
for c in sorted(acc):
    print(f"{c:5d} " + " ".join(f"{sum(acc[c][n])/len(acc[c][n])/8192:14.1f}" for n in names))
PY
