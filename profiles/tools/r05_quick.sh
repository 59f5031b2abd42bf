#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_timed_path.py -x -q -k "folded or fifteen or trajectory or real_tokenised" 2>&1 | tail -12 | cut -c1-250
bash profiles/tools/timeline_quick.sh r5h > /dev/null 2>&1; cat gpurun_out/r5h/step_timeline.txt
python3 bench.py --no-cpu-baseline --no-stress > gpurun_out/bench_lnstat.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/bench_lnstat.json')); print(d['ms_per_step'], d['window_ms_per_step'], d['kernels']['amid_sas_seq_fwd_split_lnstat_f32']['avg_launch_us'])"
