#!/bin/bash
# Like build_variant.sh, but only csrc/gemm.hip is recompiled with the extra flags; every other object is taken from the
# product build (touchnet_amd/_lib/*.o).  usage: scripts/build_gemm_variant.sh <name> [extra hipcc flags...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/touchnet_amd/_lib/variants/$name
mkdir -p "$out"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c "$root/touchnet_amd/csrc/gemm.hip" -o "$out/gemm.o"
objs=$(ls "$root"/touchnet_amd/_lib/*.o | grep -v '/gemm.o$')
hipcc -shared -fPIC --offload-arch=gfx950 $objs "$out/gemm.o" -o "$out/libtouchnet_amd.so"
rm -f "$out/gemm.o"
echo "$out/libtouchnet_amd.so"
