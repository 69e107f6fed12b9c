#!/bin/bash
# Round 6: build a variant library whose dK / dV kernels (attn_bwd.hip, attn_bwd_fused.hip) are those of commit $1 — the
# rest of the sources as they are now — into touchnet_amd/_lib/variants/$2/libtouchnet_amd.so (select with TN_AMD_LIB).
# usage: scripts/r06_build_kv_variant.sh <commit> <name>
set -e
commit=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d); cp -r "$root/touchnet_amd/csrc" "$tmp/"
git -C "$root" show "$commit:touchnet_amd/csrc/attn_bwd.hip" > "$tmp/csrc/attn_bwd.hip"
git -C "$root" show "$commit:touchnet_amd/csrc/attn_bwd_fused.hip" > "$tmp/csrc/attn_bwd_fused.hip"
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
