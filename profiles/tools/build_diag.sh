#!/bin/bash
# Diagnostic build of the whole library with in-kernel s_memtime stamps (-DAMID_STRIP_STAMPS) into profiles/tools/_diag/ (git-ignored
# like every .so; it travels to the GPU box with the snapshot).  The product library carries no stamps and none of the builds only
# A/B measurements reach (-DAMID_DIAG_VARIANTS: the four-strip N-split backward at D 128).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/profiles/tools/_diag
mkdir -p $O/obj
cd $R/amid_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DAMID_STRIP_STAMPS -DAMID_DIAG_VARIANTS $AMID_DIAG_EXTRA"
objs=""
for f in *.hip; do
  /opt/rocm/bin/hipcc $FL -c $f -o $O/obj/${f%.hip}.o &
  objs="$objs $O/obj/${f%.hip}.o"
done
for v in 3 4 5; do
  for f in bert; do
    /opt/rocm/bin/hipcc $FL -DAMID_TILE_RT=$v -Damid=amid_rt$v -DAMID_ENTRY_SUFFIX=_rt$v -c $f.hip -o $O/obj/${f}_rt$v.o &
    objs="$objs $O/obj/${f}_rt$v.o"
  done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libamid_hip_diag.so $objs
ls -la $O/libamid_hip_diag.so
