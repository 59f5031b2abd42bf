R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4px
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl -o tl -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress > $O/b.json 2>&1
cd $R
python3 profiles/tools/step_timeline.py $O/tl/tl_results.db > $O/step_timeline.txt 2>&1
rm -rf $O/tl
cut -c1-110 $O/step_timeline.txt | head -24
