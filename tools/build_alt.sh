#!/bin/bash
# Variant builds of libdetmatch_hip.so with extra -D flags for one source (A/B timing through DM_LIB_PATH):
#   bash tools/build_alt.sh <name> <source.hip> -DFLAG [-DFLAG ...]   ->  tools/altlib/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
C=detmatch_amd/csrc
mkdir -p tools/_alt tools/altlib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -DNDEBUG "$@" -c $C/$src -o tools/_alt/${name}_${src%.hip}.o
objs=""
for f in $C/*.o; do
  if [ "$(basename $f)" != "${src%.hip}.o" ]; then objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/altlib/lib_$name.so $objs tools/_alt/${name}_${src%.hip}.o
echo tools/altlib/lib_$name.so
