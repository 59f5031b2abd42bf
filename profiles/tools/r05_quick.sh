#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_seqn.py tests/test_gpu_timed_path.py -x -q -k "seqn or folded or trajectory or pieces" 2>&1 | tail -25 | cut -c1-250
for v in stamp1; do echo "== $v"; AMID_DIAG_LIB=$GRAFT_REPO_ROOT/profiles/tools/_diag/libamid_hip_$v.so python3 profiles/tools/seqn_stamps.py 2>&1 | grep -v amdgpu | cut -c1-900; done | tee gpurun_out/stamps_pipe.txt
python3 bench.py --no-cpu-baseline --no-stress > gpurun_out/bench_pipe.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/bench_pipe.json')); print(d['ms_per_step'], d['window_ms_per_step'], d['kernels']['amid_sas_seq_fwd_split_f32']['avg_launch_us'])"
