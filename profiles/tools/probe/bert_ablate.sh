#!/bin/bash
# Timing-only ablations of csrc/bert_strip.hip (results are WRONG by construction: only the step time matters): the library relinked
# with bert_strip.o compiled with -DAMID_BS_ABLATE_PHILOX (keep bits without the Philox rounds) / -DAMID_BS_ABLATE_GELU (no exp / rcp),
# into profiles/tools/_diag/ (git-ignored); run with AMID_LIB_PATH=<lib> python bench.py --model bert4rec --no-cpu-baseline
set -euo pipefail
R=$(cd "$(dirname "$0")/../../.." && pwd)
O=$R/profiles/tools/_diag
mkdir -p "$O/obj"
cd "$R/amid_amd/csrc"
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include"
others=$(ls build/*.o | grep -v '/bert_strip.o')
for v in PHILOX GELU; do
  /opt/rocm/bin/hipcc $FL -DAMID_BS_ABLATE_$v -c bert_strip.hip -o "$O/obj/bert_strip_$v.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$O/libamid_bs_$v.so" $others "$O/obj/bert_strip_$v.o"
done
ls -la "$O"/libamid_bs_*.so
