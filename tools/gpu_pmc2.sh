#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-graph --no-ramp --regions 2 > /tmp/pmc_$name.log 2>&1
  python - "/tmp/pmc_$name" <<'PY'
import sys, csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "arcle_step" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(f"  {k:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}  per-wave={sum(v)/len(v)/8192:10.2f}")
PY
}
run lvl1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM
run lvl2 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run misc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES
