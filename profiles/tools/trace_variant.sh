# step timeline of one SASRec variant (profiles/tools/variant_steps.py <name>): bash profiles/tools/trace_variant.sh itc-dr-e
R=$GRAFT_REPO_ROOT
V=${1:-itc}
mkdir -p $R/gpurun_out/tv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tv -o t -- python3 $R/profiles/tools/variant_steps.py $V > $R/gpurun_out/tv/out.txt 2> $R/gpurun_out/tv/err.log
cd $R
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/tv/t_kernel_trace.csv")))
rows = [r for r in rows if "amid" in r["Kernel_Name"] or "copyBuffer" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "pack_indices" in r["Kernel_Name"]]
i0, i1 = starts[100], starts[101]
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
print("step span us", (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3, "kernels", len(step))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {max(0, s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
PY
rm -f gpurun_out/tv/t_kernel_trace.csv
