#!/bin/bash
# PMC passes for the step kernel (separate runs, no tracing domains besides kernel-trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  rm -rf $R/gpurun_out/pmc_$name
  rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$name -o p --output-format csv -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-graph --no-ramp --regions 2 > $R/gpurun_out/pmc_$name.log 2>&1
  python - "$R/gpurun_out/pmc_$name" <<'PY'
import sys, csv, glob, collections
d=sys.argv[1]
files=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
acc=collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if "arcle_step" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(acc.items()):
    print(f"  {k:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
PY
}
run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM
run mem1 FETCH_SIZE
run mem2 WRITE_SIZE SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL
