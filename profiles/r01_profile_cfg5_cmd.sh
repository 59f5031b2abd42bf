set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/c5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5/prof -o c5 -- python3 $R/bench.py --workload cfg5-uniform --steps 20 --warmup 5 > $R/gpurun_out/c5/bench_under_prof.json 2> $R/gpurun_out/c5/prof_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c5/pmc_fetch -o f -- python3 $R/bench.py --workload cfg5-uniform --steps 4 --warmup 2 > $R/gpurun_out/c5/pmc_fetch.json 2> $R/gpurun_out/c5/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c5/pmc_write -o w -- python3 $R/bench.py --workload cfg5-uniform --steps 4 --warmup 2 > $R/gpurun_out/c5/pmc_write.json 2> $R/gpurun_out/c5/pmc_write_err.log
cd $R
rm -f gpurun_out/c5/prof/*kernel_trace.csv gpurun_out/c5/pmc_*/*kernel_trace.csv
ls -la gpurun_out/c5/*
