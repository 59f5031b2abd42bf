#!/bin/bash
cd $GRAFT_REPO_ROOT
bash profiles/tools/ab_variants.sh noslp 2>&1 | tee gpurun_out/ab_noslp.txt
