R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl -o tl -- python3 $R/bench.py --workload cfg4 --steps 480 --warmup 520 --no-cpu-baseline --no-stress $TRACE_EXTRA > /dev/null 2>&1
cd $R
python3 profiles/tools/step_timeline.py $O/tl/tl_results.db > $O/cfg4_step_timeline.txt 2>&1
rm -rf $O/tl
cat $O/cfg4_step_timeline.txt
