#!/bin/bash
# SQ counters of the step's kernels -> gpurun_out/<tag>/sq_counters.md (copy it to profiles/rNN_sq_counters.md).  ONE rocprofv3 --pmc
# pass (8 SQ slots) with --kernel-trace only, the program directly after "--".   bash profiles/tools/sq_counters.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-sq}; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/pmc -o sq -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stress --no-graph "$@" > $OUT/sq_bench.json 2> $OUT/sq_err.log
cd $R
python3 profiles/tools/sq_counters.py $OUT > $OUT/sq_counters.md
rm -rf $OUT/pmc
head -40 $OUT/sq_counters.md
