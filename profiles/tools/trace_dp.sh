# kernel timeline of ONE data-parallel step (1-rank RCCL world, graph pair + all-gather) -- bash profiles/tools/trace_dp.sh
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/trdp
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trdp -o t -- python3 $R/profiles/tools/dp_overhead.py > $R/gpurun_out/trdp/out.txt 2> $R/gpurun_out/trdp/err.log
cd $R
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/trdp/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps start with the lazy-Adam catch-up (first kernel of the local-gradients graph); take one from the fixed-bound phase
# (round 6: the folded step's first kernel is the step head)
starts = [i for i, r in enumerate(rows) if "step_head_kernel" in r["Kernel_Name"]]
i0, i1 = starts[430], starts[431]
t0 = int(rows[i0]["Start_Timestamp"])
print("step span us", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "kernels", i1 - i0)
prev = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "radix" in r["Kernel_Name"]: continue
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {max(0, s - prev) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev = max(prev, e)
PY
rm -f gpurun_out/trdp/t_kernel_trace.csv
