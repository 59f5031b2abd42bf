#!/bin/bash
# the whole GPU suite + the default bench line (round 5): gpurun -- 'bash profiles/tools/r05_full_suite.sh' -> gpurun_out/r5d/
# (+ the diagnostic library's own test -- the four-strip N-split backward against the strip build -- when that library was built,
#  and an A/B bench of profiles/tools/_diag/libamid_hip_before.so when one is there)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $O/tests_full.log
tail -4 $O/tests_full.log
D=$GRAFT_REPO_ROOT/profiles/tools/_diag
if [ -f $D/libamid_hip_diag.so ]; then
  AMID_LIB_PATH=$D/libamid_hip_diag.so timeout 600 python -m pytest tests/test_gpu_timed_path.py -x -q -k n_split 2>&1 | tail -3 | tee $O/tests_diag.log
fi
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['window_ms_per_step']); print(json.dumps(d.get('eval'))[:900]); print(d.get('cpu_baseline',{}).get('value'))"
tail -3 $O/bench.err
if [ -f $D/libamid_hip_before.so ]; then
  for i in 1 2; do
    AMID_LIB_PATH=$D/libamid_hip_before.so python3 bench.py --no-stress --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('before', d['ms_per_step'])"
    python3 bench.py --no-stress --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('now   ', d['ms_per_step'])"
  done
fi
