#!/bin/bash
# the whole GPU suite + the default bench line (round 5): bash profiles/tools/r05_full_suite.sh -> gpurun_out/r5d/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/tests_full.log
tail -4 $O/tests_full.log
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['window_ms_per_step']); print(json.dumps(d.get('eval'))[:900]); print(d.get('cpu_baseline',{}).get('value'))"
tail -3 $O/bench.err
