#!/bin/bash
# Round 4's open finding (csrc/wgrad_split.h): zeroing the rows past a split's end by a 0 / 1 MULTIPLICATION gave wrong sums in 239 of 240
# launches whenever two workgroups shared a CU; by selection it does not.  This script rebuilds that form (-DAMID_WGS_ZERO_BY_MUL) into a
# variant library (here, before the GPU run: profiles/tools/build_variant.sh), checks both forms' generated ISA with the vector-memory
# counter model (profiles/tools/probe/vmcnt_model.py: no read of a register with its load outstanding in either), and -- on the GPU box --
# runs the two-workgroups-per-CU repeat test against both libraries.
#   build (no GPU):  bash profiles/tools/probe/wgrad_opsel_repro.sh build
#   run (GPU box):   bash profiles/tools/probe/wgrad_opsel_repro.sh run      -> gpurun_out/wgrad_opsel_repro.txt
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
cd "$R"
if [ "${1:-run}" = build ]; then
  (cd amid_amd/csrc && make -j8 > /dev/null)
  bash profiles/tools/build_variant.sh zeromul "-DAMID_WGS_ZERO_BY_MUL" sasrec_wgrad_split.hip bert.hip
  for v in sel mul; do
    F=""; [ $v = mul ] && F="-DAMID_WGS_ZERO_BY_MUL"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude --cuda-device-only -S -Wno-unused-command-line-argument $F -o /tmp/wg_$v.s amid_amd/csrc/sasrec_wgrad_split.hip
    python3 profiles/tools/probe/vmcnt_model.py /tmp/wg_$v.s sas_wgrad_split_kernelILi128ELi6ELb1ELb0
  done
  exit 0
fi
O=$R/gpurun_out/wgrad_opsel_repro.txt
mkdir -p "$R/gpurun_out"
{
  echo "# product library (rows past a split's end zeroed by selection)"
  python3 -m pytest tests/test_gpu_kernels.py -q -k "two_workgroups_per_cu" 2>&1 | tail -3
  echo "# variant library (-DAMID_WGS_ZERO_BY_MUL: zeroed by a 0 / 1 multiplication)"
  AMID_LIB_PATH=$R/profiles/tools/_diag/libamid_hip_zeromul.so python3 -m pytest tests/test_gpu_kernels.py -q -k "two_workgroups_per_cu or wgrad_split_products" 2>&1 | tail -8
} > "$O" 2>&1
cat "$O"
