set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_timed_path.py -x -q -k "folded or fifteen or trajectory" 2>&1 | tail -15 > $O/tests.log
tail -5 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_sasrec.py tests/test_gpu_module.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -5 > $O/tests2.log
tail -3 $O/tests2.log
python3 profiles/tools/head_stamps.py 2>&1 | grep -v amdgpu > $O/head_stamps.txt
cat $O/head_stamps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 profiles/tools/step_timeline.py $O/tl/tl_results.db step_head > $O/step_timeline.txt 2>&1
rm -rf $O/tl
cat $O/step_timeline.txt
python3 bench.py --no-cpu-baseline --no-stress > $O/bench_new.json 2> $O/bench_new.err
python3 -c "
import json
d=json.load(open('$O/bench_new.json')); print(d['ms_per_step'], d['window_ms_per_step'], d['loss_last'])"
