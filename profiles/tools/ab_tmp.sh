cd $GRAFT_REPO_ROOT
D=$GRAFT_REPO_ROOT/profiles/tools/_diag
export AMID_LIB_PATH=$D/libamid_hip_before.so
bash profiles/tools/kstats.sh ks_before | grep -E "step_head|embed_fwd|ms/step"
unset AMID_LIB_PATH
bash profiles/tools/kstats.sh ks_now | grep -E "step_head|embed_fwd|ms/step"
