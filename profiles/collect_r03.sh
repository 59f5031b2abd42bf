#!/bin/bash
# gpurun_out/<tag>/ (scratch, written by profiles/r03_profile_cmd.sh on the MI355X box) -> the summaries committed under profiles/r03_*
set -e
TAG=${1:-r3}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
P=$R/profiles
last() { tail -n 1 "$1" > "$2"; }
last $O/bench.json $P/r03_bench.json
for w in cfg1 cfg3 cfg4 cfg3_bf16 cfg2_bf16 bert4rec whole_row_forward cfg4_steady cfg4_strip_backward; do last $O/bench_$w.json $P/r03_bench_$w.json; done
last $O/bench_cfg5-uniform.json $P/r03_bench_cfg5_uniform.json
last $O/bench_cfg5-real.json $P/r03_bench_cfg5_real.json
cp $O/prof/p_kernel_stats.csv $P/r03_bench_kernel_stats.csv
python3 $P/summarize.py stats $O/prof/p_kernel_stats.csv $P/r03_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats of the headline bench (round 3)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r03_profile_cmd.sh $TAG. The replayed step's kernels are the 15 rows with ~100+ calls; seqn_fwd_kernel = amid_sas_seq_fwd_f32 (the N-split build)."
python3 $P/summarize.py traffic $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $P/r03_cfg2_sasrec_f32_hbm_traffic.json
cp $O/sq_counters.md $P/r03_sq_counters.md
for f in step_timeline seqn_stamps seqn_bwd_stamps variant_steps dp_overhead k1_time catchup_gap d64_probe; do cp $O/$f.txt $P/r03_$f.txt; done
ls -la $P/r03_*
