#!/bin/bash
# One-box A/B of variant libraries (profiles/tools/build_variant.sh NAME ...): the headline bench once per library, interleaved with the
# product library:  bash profiles/tools/ab_variants.sh NAME [NAME ...]   -> gpurun_out/ab_<NAME>.json + one summary line each
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
mkdir -p gpurun_out
run() {   # tag, library path ("" = product)
  if [ -n "$2" ]; then export AMID_LIB_PATH=$2; else unset AMID_LIB_PATH; fi
  python3 bench.py --no-cpu-baseline --no-stress ${AB_ARGS:-} > gpurun_out/ab_$1.json 2> gpurun_out/ab_$1.err
  python3 - "$1" <<'P'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{t}.json"))
    k = d.get("kernels", {})
    f = {n: v["avg_launch_us"] for n, v in k.items()}
    top = sorted(f.items(), key=lambda x: -x[1])[:4]
    print(f"{t:12s} {d['ms_per_step']:.4f} ms/step  windows {d['window_ms_per_step']}  " + "  ".join(f"{n.replace('amid_', '')} {v}" for n, v in top))
except Exception as e:
    print(t, "ERR", e)
P
}
run product0 ""
for n in "$@"; do run "$n" "$R/profiles/tools/_diag/libamid_hip_$n.so"; done
run product1 ""
