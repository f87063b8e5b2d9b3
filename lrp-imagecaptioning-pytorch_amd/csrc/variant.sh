#!/bin/bash
# ./variant.sh <name> <source.hip> <flags...>: csrc/variants/liblrpx_<name>.so = the in-tree objects with ONE translation unit rebuilt with the
# given flags (same-box A/B of two correct builds through LRPX_LIB_PATH; a variant with experiment flags reports them in lrpx_build_flags())
set -e
cd "$(dirname "$0")"
name=$1; src=$2; shift 2
make -j8 >/dev/null
mkdir -p variants build/var_$name
obj=build/var_$name/$(basename ${src%.hip}).o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $src -o $obj
objs=$(ls build/*.o | grep -v "build/$(basename ${src%.hip}).o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o variants/liblrpx_$name.so
echo variants/liblrpx_$name.so
