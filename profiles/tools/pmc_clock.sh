#!/bin/bash
# effective clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc_clock}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o g -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline ${2:---no-graph} > $OUT/bench.json 2> $OUT/err.log
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for row in csv.DictReader(open(f[0])):
    if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k = row["Kernel_Name"][:60]
    dur = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) if "End_Timestamp" in row else 0.0
    acc[k][0] += float(row["Counter_Value"]); acc[k][1] += dur; acc[k][2] += 1
with open("$OUT/clock.txt", "w") as o:
    for k, (c, d, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        line = f"{k:60s} n={n:4d} avg {d / n / 1e3:8.2f} us  GUI_ACTIVE/8 = {c / n / 8:10.0f} cycles  clock {c / 8 / max(d, 1):.3f} GHz"
        print(line); o.write(line + "\n")
PY

rm -rf $OUT/pmc
