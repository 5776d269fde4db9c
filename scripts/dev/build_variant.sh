#!/bin/bash
# a second build of libkmx with extra -D flags, beside the product's: scripts/dev/build_variant.sh prof "-DKMX_PHASE_PROF"
# -> kmtricks_amd/libkmx_<name>.so (KMX_LIB=<path> makes kmtricks_amd/lib.py load it; test / tuning use only).
# FILES="merge_bft" recompiles only those sources and takes the product's objects for the rest.
set -e
name=$1; flags=$2
root="$(cd "$(dirname "$0")/../.." && pwd)"
obj=/tmp/kmx_objs_$name; rm -rf $obj; mkdir -p $obj
cd $root/kmtricks_amd/csrc
all="kmx_api merge_rows merge_rows_small merge_pivot merge_cols merge_cols_k2 merge_bf merge_bft count transpose superk"
for f in $all; do
  if [ -z "$FILES" ] || [[ " $FILES " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $f.hip -o $obj/$f.o &
  else cp $f.o $obj/$f.o; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/kmtricks_amd/libkmx_$name.so $obj/*.o
echo built $root/kmtricks_amd/libkmx_$name.so
