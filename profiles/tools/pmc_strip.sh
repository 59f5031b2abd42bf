#!/bin/bash
# SQ counters of the step's kernels (MFMA busy, LDS conflicts, wait states): one rocprofv3 --pmc pass, program directly after --
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_strip}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc -o sq -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-graph > $OUT/bench.json 2> $OUT/err.log
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
print(f)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:70]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
with open("$OUT/summary.txt", "w") as o:
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
        m = max(n[k], 1)
        line = f"{k:70s} n={m:4d} " + " ".join(f"{cn[3:]}={v / m:.3g}" for cn, v in sorted(c.items()))
        print(line); o.write(line + "\n")
PY
rm -rf $OUT/pmc
