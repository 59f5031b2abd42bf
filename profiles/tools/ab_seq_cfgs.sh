# fused per-sequence backward on / off at the other configurations (one box):  gpurun -- 'bash profiles/tools/ab_seq_cfgs.sh'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rm -f $O/b_*.json
for w in cfg2 cfg3 cfg4 cfg5-real; do
  for s in 1 0; do
    timeout 300 python3 bench.py --set SEQ_BACKWARD=$s --workload $w --no-stress --no-cpu-baseline > $O/b_${w}_seq$s.json 2> $O/b_${w}_seq$s.err
  done
done
python3 profiles/tools/_ab.py
