#!/bin/bash
# Build an experimental variant of libtouchnet_amd.so with extra hipcc flags (e.g. -DTN_PP_EXP) into
# touchnet_amd/_lib/variants/<name>/libtouchnet_amd.so; select it at run time with TN_AMD_LIB=<path>.
# usage: scripts/build_variant.sh <name> [extra hipcc flags...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/touchnet_amd/_lib/variants/$name
mkdir -p "$out"
pids=()
for src in "$root"/touchnet_amd/csrc/*.hip; do
  extra=""; case "$(basename "$src")" in attn_fwd.hip|attn_fwd_pp.hip) extra="-fno-honor-nans";; esac   # = build.py EXTRA_FLAGS
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $extra "$@" -c "$src" -o "$out/$(basename "${src%.hip}").o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc -shared -fPIC --offload-arch=gfx950 "$out"/*.o -o "$out/libtouchnet_amd.so"
rm -f "$out"/*.o
echo "$out/libtouchnet_amd.so"
