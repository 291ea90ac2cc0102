#!/bin/bash
# Builds schwarzwald_amd/lib/libswz_v<name>.so: the library with ONE translation unit recompiled with extra flags
# (experiments; tools/probe.sh variants times them through SWZ_GPU_LIBRARY).  usage: build_variant.sh <name> <file.hip> <flags...>
set -euo pipefail
cd "$(dirname "$0")/../schwarzwald_amd/csrc"
name=$1; src=$2; shift 2
make -s -j8 >/dev/null
mkdir -p build_v
obj=build_v/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" -c $src -o $obj
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libswz_v$name.so $objs $obj -lz
echo built ../lib/libswz_v$name.so
