set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5b
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_timed_path.py -x -q -k "folded or fifteen or trajectory" 2>&1 | tail -15 > $O/tests.log
tail -5 $O/tests.log
timeout 600 python -m pytest tests/test_gpu_sasrec.py -x -q -k "pool" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 profiles/tools/step_timeline.py $O/tl/tl_results.db step_head > $O/step_timeline.txt 2>&1
rm -rf $O/tl
cat $O/step_timeline.txt
