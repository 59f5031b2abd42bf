set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_gpu_timed_path.py -x -q -k "folded or fifteen or trajectory" 2>&1 | tail -15 > gpurun_out/r5a/tests.log
tail -5 gpurun_out/r5a/tests.log
python3 bench.py --no-cpu-baseline --no-stress > gpurun_out/r5a/bench_new.json 2> gpurun_out/r5a/bench_new.err
python3 bench.py --no-cpu-baseline --no-stress --no-fused-tail > gpurun_out/r5a/bench_old.json 2> gpurun_out/r5a/bench_old.err
python3 bench.py --no-cpu-baseline --no-stress > gpurun_out/r5a/bench_new2.json 2> gpurun_out/r5a/bench_new2.err
python3 -c "
import json
for n in ('new','old','new2'):
    try:
        d=json.load(open('gpurun_out/r5a/bench_%s.json'%n)); print(n, d['ms_per_step'], d['window_ms_per_step'], d['loss_last'])
    except Exception as e: print(n, 'ERR', e)
"
tail -3 gpurun_out/r5a/bench_new.err
