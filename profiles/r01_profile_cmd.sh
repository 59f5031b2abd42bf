set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r1d
python3 $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/r1d/bench.json 2> $R/gpurun_out/r1d/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r1d/prof -o r1d -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r1d/bench_under_prof.json 2> $R/gpurun_out/r1d/prof_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r1d/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1d/pmc_fetch.json 2> $R/gpurun_out/r1d/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r1d/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r1d/pmc_write.json 2> $R/gpurun_out/r1d/pmc_write_err.log
cd $R
rm -f gpurun_out/r1d/prof/*kernel_trace.csv gpurun_out/r1d/pmc_*/*kernel_trace.csv
ls -la gpurun_out/r1d gpurun_out/r1d/*
tail -c 600 gpurun_out/r1d/bench.json
