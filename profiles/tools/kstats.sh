# Per-kernel average durations of the replayed step: bash profiles/tools/kstats.sh <tag> [bench.py arguments]
# (rocprofv3 --kernel-trace --stats of bench.py, the program itself behind "--"; prints the 16 kernels with the most total time)
TAG=${1:-ks}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress "$@" > $O/bench_under_prof.json 2> $O/prof_err.log
cd $R
rm -f $O/prof/*kernel_trace.csv
python3 - $O <<'PY'
import csv, glob, json, sys
o = sys.argv[1]
f = glob.glob(o + "/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:16]
for r in rows:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:8.2f} us')
d = json.loads(open(o + "/bench_under_prof.json").read().strip().splitlines()[-1])
print("ms/step under the profiler:", d["ms_per_step"])
PY
