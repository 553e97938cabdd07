#!/bin/bash
# instruction counts per step-kernel instantiation of any command (default: the research-env bench), one PMC pass:
#   bash tools/gpu_kpmc.sh python tools/researchbench.py
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_k
if [ $# -eq 0 ]; then set -- python $R/tools/researchbench.py; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --kernel-trace -d $R/gpurun_out/pmc_k -o p --output-format csv -- "$@" > /dev/null 2>&1
python - $R/gpurun_out/pmc_k <<'PY'
import sys, csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
grid={}
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "arcle_" in row["Kernel_Name"]:
            k=row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            grid[k]=int(row.get("Grid_Size",0) or 0)//64
names=sorted({n for k in acc for n in acc[k]})
print(f"{'kernel':60s} waves " + " ".join(f"{n.replace('SQ_',''):>12s}" for n in names) + "   (per wave)")
for k in sorted(acc):
    wv=max(grid[k],1)
    print(f"{k[-60:]:60s} {wv:5d} " + " ".join(f"{sum(acc[k][n])/len(acc[k][n])/wv:12.1f}" for n in names))
PY
