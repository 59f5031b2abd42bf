#!/bin/bash
# A variant of the product library for one-box A/B runs: build_variant.sh NAME "-DFLAG=..." file.hip [file.hip ...] recompiles the named
# sources with the extra flags, links them with the product build's other objects (amid_amd/csrc/build/, `make` first) into
# profiles/tools/_diag/libamid_hip_NAME.so (git-ignored like every .so; it travels to the GPU box).  Use: AMID_LIB_PATH=<that file>.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; FLAGS=$2; shift 2
O=$R/profiles/tools/_diag/var_$NAME
mkdir -p $O
cd $R/amid_amd/csrc
objs=""
for o in build/*.o; do
  b=$(basename $o .o); skip=0
  for f in "$@"; do [ "${f%.hip}" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $FLAGS -c $f -o $O/${f%.hip}.o &
  objs="$objs $O/${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/profiles/tools/_diag/libamid_hip_$NAME.so $objs
ls -la $R/profiles/tools/_diag/libamid_hip_$NAME.so
