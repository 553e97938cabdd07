#!/bin/bash
# instruction counts of the three ingress forms (tools/maskbench.py under a PMC pass), per kernel instantiation
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mask
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH --kernel-trace -d $R/gpurun_out/pmc_mask -o p --output-format csv -- python $R/tools/maskbench.py > /dev/null 2>&1
python - $R/gpurun_out/pmc_mask <<'PY'
import sys, csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "arcle_step" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
names=sorted({n for k in acc for n in acc[k]})
print(f"{'kernel':44s} " + " ".join(f"{n.replace('SQ_',''):>14s}" for n in names) + "   (per wave)")
for k in sorted(acc):
    print(f"{k[-44:]:44s} " + " ".join(f"{sum(acc[k][n])/len(acc[k][n])/8192:14.1f}" for n in names))
PY
