#!/bin/bash
# gpurun_out/<tag>/ (scratch, written by profiles/r05_profile_cmd.sh on the MI355X box) -> the summaries committed under profiles/r05_*
set -euo pipefail
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
P=$R/profiles
last() { [ -s "$1" ] || { echo "collect_r05: $1 is missing or empty -- nothing copied" >&2; exit 1; }; tail -n 1 "$1" > "$2"; }
last $O/bench.json $P/r05_bench.json
last $O/bench_driver_args.json $P/r05_bench_driver_args.json
for w in cfg1 cfg3 cfg4 cfg3_bf16 cfg2_bf16 bert4rec bert4rec_fp32_strips cfg4_steady fifteen_launches twelve_launches fp32_wgrad fp32_forward fp32_bwd_strips; do last $O/bench_$w.json $P/r05_bench_$w.json; done
last $O/bench_cfg5-uniform.json $P/r05_bench_cfg5_uniform.json
last $O/bench_cfg5-real.json $P/r05_bench_cfg5_real.json
cp $O/prof/p_kernel_stats.csv $P/r05_bench_kernel_stats.csv
python3 $P/summarize.py stats $O/prof/p_kernel_stats.csv $P/r05_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats of the headline bench (round 5)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r05_profile_cmd.sh $TAG. The replayed step's kernels are the 11 rows with ~100+ calls (attn_bwd_mfma_kernel runs twice a step): step_head_kernel (amid_step_head_f32), embed_fwd_kernel (K1), seqn_fwd_px_head_kernel (amid_sas_seq_fwd_split_lnstat_head_f32: the forward with the step's head on the tail of its workgroups), strip_ffn_bwd_kernel, attn_bwd_mfma_kernel x 2, strip_qkv_bwd_kernel<..., true, ...> (+ sort and scorer riders) and <..., false, ...> (+ the embedding epilogue), sas_wgrad_split_kernel (+ sort phase 5, LayerNorm rebuilt), grad_tail_live_kernel, optimizer_step_spans_kernel."
cp $O/prof_bert/p_kernel_stats.csv $P/r05_bench_bert4rec_kernel_stats.csv
python3 $P/summarize.py stats $O/prof_bert/p_kernel_stats.csv $P/r05_bench_bert4rec_kernel_stats.md "rocprofv3 --kernel-trace --stats of the BERT4Rec bench (round 5)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof_bert -o p -- python3 bench.py --model bert4rec --steps 100 --warmup 10 --no-cpu-baseline --no-stress" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r05_profile_cmd.sh $TAG."
python3 $P/summarize.py traffic $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $P/r05_cfg2_sasrec_f32_hbm_traffic.json
for f in step_timeline step_timeline_fifteen_launches step_timeline_twelve_launches head_stamps bert_step_timeline seqn_stamps variant_steps dp_overhead k1_time strip_bwd_stamps; do [ -f $O/$f.txt ] && cp $O/$f.txt $P/r05_$f.txt; done
[ -f $O/sq_counters.md ] && cp $O/sq_counters.md $P/r05_sq_counters.md
ls -la $P/r05_*
