# A/B of the fused per-sequence backward on one box: timed-path parity tests, bench with --set SEQ_BACKWARD=auto / 0, step timeline of the
# fused configuration.   gpurun -- 'bash profiles/tools/ab_seq.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 400 python3 -m pytest tests/test_gpu_timed_path.py -q -x > $O/tp1.log 2>&1; tail -2 $O/tp1.log
rm -f $O/b_*.json
timeout 200 python3 bench.py --no-stress --no-cpu-baseline > $O/b_seq1.json 2> $O/b_seq1.err
timeout 200 python3 bench.py --set SEQ_BACKWARD=0 --no-stress --no-cpu-baseline > $O/b_seq0.json 2> $O/b_seq0.err
python3 profiles/tools/_ab.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/tl1 -o tl -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress > /dev/null 2>&1
cd $R
python3 profiles/tools/step_timeline.py $(find gpurun_out/tl1 -name "*.db" | head -1) > $O/tl_seq1.txt 2>&1
rm -rf $O/tl1
cat $O/tl_seq1.txt
