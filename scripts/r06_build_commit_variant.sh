#!/bin/bash
# Round 6: build the kernel sources of commit $1 as a variant library touchnet_amd/_lib/variants/$2/libtouchnet_amd.so
# (select with TN_AMD_LIB; the Python side stays the working tree's, so the ABI of the two must agree).
# usage: scripts/r06_build_commit_variant.sh <commit> <name>
set -e
commit=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d); mkdir -p "$tmp/csrc"
for f in $(git -C "$root" ls-tree --name-only "$commit" touchnet_amd/csrc/); do
  git -C "$root" show "$commit:$f" > "$tmp/csrc/$(basename $f)"
done
out=$root/touchnet_amd/_lib/variants/$name; mkdir -p "$out"
pids=()
for src in "$tmp"/csrc/*.hip; do
  extra=""; case "$(basename "$src")" in attn_fwd.hip|attn_fwd_pp.hip|attn_fwd_stream.hip) extra="-fno-honor-nans";; esac
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result $extra -c "$src" -o "$out/$(basename "${src%.hip}").o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc -shared -fPIC --offload-arch=gfx950 "$out"/*.o -o "$out/libtouchnet_amd.so"
rm -f "$out"/*.o; rm -rf "$tmp"
echo "$out/libtouchnet_amd.so"
