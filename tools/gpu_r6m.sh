#!/bin/bash
# Round 6, call M: the big-grid step kernel under FULL load (16 384 envs, eager launches): wave lifetime, wait and active cycles per wave
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for size in 40x40 64x64; do for envs in 4096 16384; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_LDS" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
    rm -rf $O/pmc_m
    rocprofv3 --pmc $set --kernel-trace -d $O/pmc_m -o p --output-format csv -- python $R/tools/bigbench.py --eager --sizes $size --envs $envs --steps 12 > /dev/null 2>&1
    python - $O/pmc_m "$size $envs" <<'PY'
import sys, csv, glob, collections
acc=collections.defaultdict(list); grid=0
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "big_step" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"])); grid=int(row.get("Grid_Size",0) or 0)//64
print(sys.argv[2], "waves", grid, " ".join(f"{k.replace('SQ_','').replace('SQC_','C_')}={sum(v)/len(v)/max(grid,1):.1f}" for k,v in sorted(acc.items())))
PY
  done
done; done | tee $O/r6m_load.txt
