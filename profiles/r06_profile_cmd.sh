#!/bin/bash
# Round-6 evidence run (one MI355X box): bash profiles/r06_profile_cmd.sh <tag>   -> gpurun_out/<tag>/..., summarised into profiles/ by
# profiles/collect_r06.sh (see profiles/README.md).  Every rocprofv3 command has the program itself behind "--"; counters are collected
# in their own passes (--kernel-trace only).  A/B lines set a class attribute of the engine (bench.py --set NAME=VALUE).
set -uo pipefail
set -x
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
# the driver's own command line, then the default one
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
python3 bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress > $O/bench_under_prof.json 2> $O/prof_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stress > $O/pmc_fetch.json 2> $O/pmc_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stress > $O/pmc_write.json 2> $O/pmc_write_err.log
rocprofv3 --kernel-trace -d $O/tl -o tl -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress > /dev/null 2>&1
rocprofv3 --kernel-trace -d $O/tlb -o tl -- python3 $R/bench.py --model bert4rec --steps 60 --warmup 10 --no-cpu-baseline --no-stress > /dev/null 2>&1
rocprofv3 --kernel-trace -d $O/tl5 -o tl -- python3 $R/bench.py --set GATHER_ON_FWD=0 --steps 60 --warmup 10 --no-cpu-baseline --no-stress > /dev/null 2>&1
# the evaluation loop alone: kernel stats and HBM traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eval -o e -- python3 $R/profiles/tools/eval_bench.py 256 > $O/eval_under_prof.json 2> $O/prof_eval_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_eval_fetch -o f -- python3 $R/profiles/tools/eval_bench.py 16 > /dev/null 2> $O/pmc_eval_fetch_err.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_eval_write -o w -- python3 $R/profiles/tools/eval_bench.py 16 > /dev/null 2> $O/pmc_eval_write_err.log
cd "$R"
rm -f $O/prof/*kernel_trace.csv $O/pmc_*/*kernel_trace.csv $O/prof_eval/*kernel_trace.csv
python3 profiles/tools/step_timeline.py $O/tl/tl_results.db > $O/step_timeline.txt 2>&1
python3 profiles/tools/step_timeline.py $O/tlb/tl_results.db > $O/bert_step_timeline.txt 2>&1
python3 profiles/tools/step_timeline.py $O/tl5/tl_results.db > $O/step_timeline_ten_launches.txt 2>&1
rm -rf $O/tl $O/tlb $O/tl5
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
for w in cfg5-uniform cfg5-real cfg4 cfg3 cfg1; do python3 bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; done
python3 bench.py --model bert4rec --no-cpu-baseline > $O/bench_bert4rec.json 2> $O/bench_bert4rec.err
python3 bench.py --dtype bf16 --no-cpu-baseline > $O/bench_cfg2_bf16.json 2> $O/bench_cfg2_bf16.err
python3 bench.py --workload cfg3 --dtype bf16 --no-cpu-baseline > $O/bench_cfg3_bf16.json 2> $O/bench_cfg3_bf16.err
# the bf16 step as it ran before it folded (round 5's launches with bf16 products)
python3 bench.py --dtype bf16 --set BF16_FOLD=0 --no-cpu-baseline --no-stress > $O/bench_cfg2_bf16_unfolded.json 2> $O/bench_cfg2_bf16_unfolded.err
python3 bench.py --workload cfg3 --dtype bf16 --set BF16_FOLD=0 --no-cpu-baseline --no-stress > $O/bench_cfg3_bf16_unfolded.json 2> $O/bench_cfg3_bf16_unfolded.err
# A/B on this box: the embedding gather as its own launch (ten launches); the optimizer as its own launch too (round 5's eleven launches);
# the head as its own launch as well; round 4's fifteen launches
python3 bench.py --set GATHER_ON_FWD=0 --no-cpu-baseline --no-stress > $O/bench_ten_launches.json 2> $O/bench_ten_launches.err
python3 bench.py --set GATHER_ON_FWD=0 --set FUSED_OPT=0 --no-cpu-baseline --no-stress > $O/bench_eleven_launches.json 2> $O/bench_eleven_launches.err
python3 bench.py --set GATHER_ON_FWD=0 --set FUSED_OPT=0 --set HEAD_ON_FWD=0 --no-cpu-baseline --no-stress > $O/bench_twelve_launches.json 2> $O/bench_twelve_launches.err
python3 bench.py --no-fused-tail --no-cpu-baseline --no-stress > $O/bench_fifteen_launches.json 2> $O/bench_fifteen_launches.err
(echo "# python profiles/tools/variant_steps.py (cfg 2 shape: B 256, T 50, D 128, hid 32, neg 1; hipGraph replay, 200 steps)"; python3 profiles/tools/variant_steps.py 2>&1 | grep "ms/step"; echo "# VARIANT_T=20 (the mybank shape run.sh trains on)"; VARIANT_T=20 python3 profiles/tools/variant_steps.py 2>&1 | grep "ms/step") > $O/variant_steps.txt
python3 profiles/tools/dp_overhead.py 2>&1 | grep "ms/step" > $O/dp_overhead.txt
bash profiles/tools/trace_dp.sh > $O/dp_step_timeline.txt 2>&1
python3 profiles/tools/k1_time.py 2>&1 | grep "TB/s" > $O/k1_time.txt
python3 profiles/tools/cli_throughput.py --cfg1 2>&1 | grep -E "^cfg 1|^wall" > $O/cli_cfg1.txt
python3 profiles/tools/cli_throughput.py --dr 2>&1 | grep -E "samples/s|^wall" | tail -6 > $O/cli_runsh.txt
bash profiles/tools/sq_counters.sh $TAG > /dev/null 2>&1
python3 bench.py --workload cfg4 --steps 480 --warmup 520 --no-cpu-baseline --no-stress > $O/bench_cfg4_steady.json 2> $O/bench_cfg4_steady.err
# cfg 4 (T 20): the one-launch backward with its side-stream sort instead of the folded strips; ... with the whole sort chained in the catch-up launch;
# BERT4Rec with the chained sort; one cfg 4 step kernel by kernel
python3 bench.py --workload cfg4 --steps 480 --warmup 520 --set FOLD_SHORT=0 --no-cpu-baseline --no-stress > $O/bench_cfg4_steady_unfolded.json 2> $O/bench_cfg4_steady_unfolded.err
python3 bench.py --workload cfg4 --steps 480 --warmup 520 --set FOLD_SHORT=0 --set SORT_CHAIN=1 --no-cpu-baseline --no-stress > $O/bench_cfg4_steady_chain.json 2> $O/bench_cfg4_steady_chain.err
python3 bench.py --model bert4rec --set SORT_CHAIN=1 --no-cpu-baseline --no-stress > $O/bench_bert4rec_chain.json 2> $O/bench_bert4rec_chain.err
bash profiles/tools/trace_cfg4.sh > /dev/null 2>&1; cp $R/gpurun_out/tl4/cfg4_step_timeline.txt $O/cfg4_step_timeline.txt
ls -la $O
tail -3 $O/smoke.log
tail -c 600 $O/bench_driver_args.json
