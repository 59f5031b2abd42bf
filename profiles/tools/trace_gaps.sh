R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tr
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr -o t -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/tr/out.json 2> $R/gpurun_out/tr/err.log
cd $R
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open("gpurun_out/tr/t_kernel_trace.csv")))
rows = [r for r in rows if "amid" in r["Kernel_Name"] or "copyBuffer" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find graph-replay steps: sequences starting with pack_indices; take step #30 (well inside the timed region)
starts = [i for i, r in enumerate(rows) if "pack_indices" in r["Kernel_Name"]]
i0, i1 = starts[30], starts[31]
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
print("step span us", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "kernels", len(step))
busy = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {max(0, s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:60]}")
    prev_end = max(prev_end, e)
PY
rm -f gpurun_out/tr/t_kernel_trace.csv
