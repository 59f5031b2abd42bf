#!/bin/bash
# round-5 working run -> gpurun_out/r5f/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_timed_path.py -x -q -k "folded or trajectory" 2>&1 | tail -3
(python3 profiles/tools/probe/wgrad_opsel_probe.py; AMID_LIB_PATH=$GRAFT_REPO_ROOT/profiles/tools/_diag/libamid_hip_zeromul.so python3 profiles/tools/probe/wgrad_opsel_probe.py) 2>&1 | grep -v amdgpu | tee $O/wgrad_opsel_probe.txt
bash profiles/tools/timeline_quick.sh r5f > /dev/null 2>&1; cat gpurun_out/r5f/step_timeline.txt
python3 bench.py --no-cpu-baseline --no-stress > $O/bench.json 2>$O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['window_ms_per_step'])"
