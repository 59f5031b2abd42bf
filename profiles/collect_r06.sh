#!/bin/bash
# gpurun_out/<tag>/ (scratch, written by profiles/r06_profile_cmd.sh on the MI355X box) -> the summaries committed under profiles/r06_*
set -euo pipefail
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
P=$R/profiles
last() { [ -s "$1" ] || { echo "collect_r06: $1 is missing or empty -- nothing copied" >&2; exit 1; }; tail -n 1 "$1" > "$2"; }
last $O/bench.json $P/r06_bench.json
last $O/bench_driver_args.json $P/r06_bench_driver_args.json
for w in cfg1 cfg3 cfg4 cfg3_bf16 cfg2_bf16 cfg3_bf16_unfolded cfg2_bf16_unfolded bert4rec cfg4_steady cfg4_steady_unfolded cfg4_steady_chain bert4rec_chain ten_launches eleven_launches twelve_launches fifteen_launches; do last $O/bench_$w.json $P/r06_bench_$w.json; done
last $O/bench_cfg5-uniform.json $P/r06_bench_cfg5_uniform.json
last $O/bench_cfg5-real.json $P/r06_bench_cfg5_real.json
last $O/eval_under_prof.json $P/r06_eval_under_prof.json
cp $O/prof/p_kernel_stats.csv $P/r06_bench_kernel_stats.csv
python3 $P/summarize.py stats $O/prof/p_kernel_stats.csv $P/r06_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats of the headline bench (round 6)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -o p -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r06_profile_cmd.sh $TAG. The replayed step's kernels are the 9 rows with ~100+ calls (attn_bwd_mfma_kernel runs twice a step): step_head_kernel<true> (amid_step_head_w16_f32: index marshal, lazy-Adam catch-up, the forward's bf16 weight images), seqn_fwd_px_head_kernel (amid_sas_seq_fwd_gather_head_f32: the forward with the embedding gather in its prologue and the step's head on the tail of its workgroups), strip_ffn_bwd_kernel, attn_bwd_mfma_kernel x 2, strip_qkv_bwd_kernel<..., true, ...> (+ sort and scorer riders) and <..., false, ...> (+ the embedding epilogue), sas_wgrad_split_kernel (+ sort phase 5, LayerNorm rebuilt), grad_tail_opt_kernel<2, true> (amid_grad_tail_opt_f32: the gradient tail with the optimizer inside)."
cp $O/prof_eval/e_kernel_stats.csv $P/r06_eval_kernel_stats.csv
python3 $P/summarize.py stats $O/prof_eval/e_kernel_stats.csv $P/r06_eval_kernel_stats.md "rocprofv3 --kernel-trace --stats of the evaluation loop (round 6)" \
  "rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof_eval -o e -- python3 profiles/tools/eval_bench.py 256" \
  "One MI355X box, ROCm 7.2; produced by bash profiles/r06_profile_cmd.sh $TAG. test() at 999 negatives, batch 256 x seq 50 x dim 128: per batch THREE launches in a replayed graph (the eval head's scorer chains in packed fp32 FMAs since the round's last commit) -- pack_indices_kernel (index marshal + live list), seqn_fwd_px_kernel (amid_sas_seq_fwd_gather_infer_f32: the inference forward, the own-domain sequences' rows gathered in its prologue, nothing saved), eval_head_fast_kernel (amid_eval_head_f32: LN_last + mean, the 1 000 candidates gathered inside the scorer, masked BCE, ranks) -- between two device copies (the packed batch in, 3 B result words out)."
python3 $P/summarize.py traffic $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $P/r06_cfg2_sasrec_f32_hbm_traffic.json
python3 $P/summarize.py traffic $O/pmc_eval_fetch/f_counter_collection.csv $O/pmc_eval_write/w_counter_collection.csv $P/r06_eval_hbm_traffic.json
for f in step_timeline step_timeline_ten_launches cfg4_step_timeline bert_step_timeline variant_steps dp_overhead dp_step_timeline k1_time cli_cfg1 cli_runsh; do [ -f $O/$f.txt ] && cp $O/$f.txt $P/r06_$f.txt; done
[ -f $O/sq_counters.md ] && cp $O/sq_counters.md $P/r06_sq_counters.md
ls -la $P/r06_*
