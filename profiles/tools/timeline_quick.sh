#!/bin/bash
# One replayed step's kernel timeline (start, duration, workgroups, LDS) of the headline bench under rocprofv3 --kernel-trace:
#   bash profiles/tools/timeline_quick.sh [tag] [bench.py arguments ...]   -> gpurun_out/<tag>/step_timeline.txt
# The program itself follows "--" (no env / shell hop: the profiler's preloaded library initialises the GPU before the program starts).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${1:-tl}
shift || true
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$O/tl" -o tl -- python3 "$R/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-stress "$@" > "$O/b.json" 2> "$O/b.err"
cd "$R"
ANCHOR=step_head
case " $* " in *" --no-fused-tail "*|*"bert4rec"*|*"--dtype bf16"*) ANCHOR=pack_indices;; esac
python3 profiles/tools/step_timeline.py "$O/tl/tl_results.db" $ANCHOR > "$O/step_timeline.txt" 2>&1 || python3 profiles/tools/step_timeline.py "$O/tl/tl_results.db" pack_indices > "$O/step_timeline.txt" 2>&1
rm -rf "$O/tl"
cat "$O/step_timeline.txt"
