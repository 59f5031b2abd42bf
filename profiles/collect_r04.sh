#!/bin/bash
# gpurun_out/<tag>/ (scratch, written by profiles/r04_profile_cmd.sh on the MI355X box) -> the summaries committed under profiles/r04_*
set -euo pipefail
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
P=$R/profiles
last() { tail -n 1 "$1" > "$2"; }
last $O/bench.json $P/r04_bench.json
last $O/bench_driver_args.json $P/r04_bench_driver_args.json
for w in cfg1 cfg3 cfg4 cfg3_bf16 cfg2_bf16 bert4rec cfg4_steady fp32_wgrad bert4rec_fp32_wgrad fold_catchup fp32_forward fp32_bwd_strips fp32_everything; do last $O/bench_$w.json $P/r04_bench_$w.json; done
last $O/bench_cfg5-uniform.json $P/r04_bench_cfg5_uniform.json
last $O/bench_cfg5-real.json $P/r04_bench_cfg5_real.json
cp $O/prof/p_kernel_stats.csv $P/r04_bench_kernel_stats.csv
python3 $P/summarize.py stats $O/prof/p_kernel_stats.csv $P/r04_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats of the headline bench (round 4)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r04_profile_cmd.sh $TAG. The replayed step's kernels are the 15 rows with ~100+ calls; seqn_fwd_px_kernel = amid_sas_seq_fwd_split_f32 (the one-launch forward, products on bf16 pieces made by the operand's producer), strip_*_bwd_kernel<..., 3> = the backward strips on bf16 pieces, sas_wgrad_split_kernel = the weight gradients on bf16 pieces."
cp $O/prof_bert/p_kernel_stats.csv $P/r04_bench_bert4rec_kernel_stats.csv
python3 $P/summarize.py stats $O/prof_bert/p_kernel_stats.csv $P/r04_bench_bert4rec_kernel_stats.md "rocprofv3 --kernel-trace --stats of the BERT4Rec bench (round 4)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof_bert -o p -- python3 bench.py --model bert4rec --steps 100 --warmup 10 --no-cpu-baseline --no-stress" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r04_profile_cmd.sh $TAG."
python3 $P/summarize.py traffic $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $P/r04_cfg2_sasrec_f32_hbm_traffic.json
for f in step_timeline bert_step_timeline seqn_stamps variant_steps variant_kernels dp_overhead k1_time wgrad_split_probe strip_bwd_stamps; do cp $O/$f.txt $P/r04_$f.txt; done
ls -la $P/r04_*
