R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|SQC_INST|SQC_DCACHE" | head -20
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY"; do
rm -rf /tmp/pmc_ic
rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_ic -o p --output-format csv -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > /tmp/pmc_ic.log 2>&1
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_ic/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "arcle_step" in row.get("Kernel_Name",""):
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"  {k:32s} mean={sum(v)/len(v):12.1f} per-wave={sum(v)/len(v)/8192:8.2f}")
PY
done
