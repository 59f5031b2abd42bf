#!/bin/bash
cd $GRAFT_REPO_ROOT
AMID_LIB_PATH=$GRAFT_REPO_ROOT/profiles/tools/_diag/libamid_hip_ringearly.so timeout 900 python -m pytest tests/test_gpu_seqn.py -x -q 2>&1 | tail -2
for v in stamp0 stamp1 stamp0 stamp1; do echo "== $v"; AMID_DIAG_LIB=$GRAFT_REPO_ROOT/profiles/tools/_diag/libamid_hip_$v.so python3 profiles/tools/seqn_stamps.py 2>&1 | grep -v amdgpu | cut -c1-900; done | tee gpurun_out/stamps_ringearly.txt
bash profiles/tools/ab_variants.sh ringearly
