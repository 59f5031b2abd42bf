R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc2 -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc2/out.json 2> $R/gpurun_out/pmc2/err.log
cd $R; rm -f gpurun_out/pmc2/p_kernel_trace.csv; ls gpurun_out/pmc2; tail -3 gpurun_out/pmc2/err.log
