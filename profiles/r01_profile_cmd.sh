set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1h
python3 $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/r1h/bench.json 2> $R/gpurun_out/r1h/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1h/prof -o r1h -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r1h/bench_under_prof.json 2> $R/gpurun_out/r1h/prof_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r1h/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1h/pmc_fetch.json 2> $R/gpurun_out/r1h/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r1h/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1h/pmc_write.json 2> $R/gpurun_out/r1h/pmc_write_err.log
cd $R
rm -f gpurun_out/r1h/prof/*kernel_trace.csv gpurun_out/r1h/pmc_*/*kernel_trace.csv
ls -la gpurun_out/r1h gpurun_out/r1h/*
tail -c 600 gpurun_out/r1h/bench.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r1h/smoke.log 2>&1
for w in cfg5-uniform cfg5-real cfg4; do python3 bench.py --workload $w --no-cpu-baseline > gpurun_out/r1h/bench_$w.json 2> gpurun_out/r1h/bench_$w.err; done
python3 bench.py --model bert4rec --no-cpu-baseline > gpurun_out/r1h/bench_bert4rec.json 2> gpurun_out/r1h/bench_bert4rec.err
python3 bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/r1h/bench_cfg2_bf16.json 2> gpurun_out/r1h/bench_cfg2_bf16.err
python3 bench.py --workload cfg3 --no-cpu-baseline > gpurun_out/r1h/bench_cfg3_f32.json 2> gpurun_out/r1h/bench_cfg3_f32.err
python3 bench.py --workload cfg3 --dtype bf16 --no-cpu-baseline > gpurun_out/r1h/bench_cfg3_bf16.json 2> gpurun_out/r1h/bench_cfg3_bf16.err
python3 profiles/tools/dp_overhead.py 2>&1 | grep "dp path" > gpurun_out/r1h/dp_overhead.txt
bash profiles/tools/trace_gaps.sh > gpurun_out/r1h/step_timeline.txt 2>&1
tail -3 gpurun_out/r1h/smoke.log
