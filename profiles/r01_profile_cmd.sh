set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1f
python3 $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/r1f/bench.json 2> $R/gpurun_out/r1f/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1f/prof -o r1f -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r1f/bench_under_prof.json 2> $R/gpurun_out/r1f/prof_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r1f/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1f/pmc_fetch.json 2> $R/gpurun_out/r1f/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r1f/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1f/pmc_write.json 2> $R/gpurun_out/r1f/pmc_write_err.log
cd $R
rm -f gpurun_out/r1f/prof/*kernel_trace.csv gpurun_out/r1f/pmc_*/*kernel_trace.csv
ls -la gpurun_out/r1f gpurun_out/r1f/*
tail -c 600 gpurun_out/r1f/bench.json
