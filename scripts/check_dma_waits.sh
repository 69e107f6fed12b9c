#!/bin/bash
# hipcc's waitcnt insertion puts `s_waitcnt vmcnt(0)` in front of an LDS read that may alias a pending LDS-DMA when the
# read carries no alias metadata (attn_common.h, i32x4_t).  That silently turns the counted-vmcnt stage ring of the
# attention backward kernels into a serial load -> compute loop.  This check compiles attn_bwd.hip to ISA and fails if a
# `s_waitcnt vmcnt(0)` sits directly in front of a ds_read_b128 (the MFMA "row" operand / statistics reads) inside the
# stage loops of the dK/dV and dQ kernels.  (The transpose-read builtin still gets such a wait; at the ring depth in use,
# one stage, it costs nothing — see DESIGN.md 5.2.)
# usage: scripts/check_dma_waits.sh   -> exit 0 / 1
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
for f in attn_bwd attn_bwd_dq_stream attn_fwd_stream; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans -Wno-unused-result -S --cuda-device-only \
    "$root/touchnet_amd/csrc/$f.hip" -o "$tmp/$f.s" 2>/dev/null
done
cat "$tmp"/attn_bwd.s "$tmp"/attn_bwd_dq_stream.s "$tmp"/attn_fwd_stream.s > "$tmp/bwd.s"
python3 - "$tmp/bwd.s" <<'PY'
import re, sys
text = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r'^(_ZN2tn\d+attn_(?:bwd_kv|bwd_dq|bwd_dq_stream|fwd_stream)_kernel\S*):.*?s_endpgm', text, re.S | re.M):
    name, body = m.group(1), m.group(0).split('\n')
    if 'fwd_stream_kernelILi128ELb1E' in name:      # the TRACE instantiation (s_memtime stamps + stores: development only)
        continue
    ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((';', '.'))]
    # stage loop = everything after the first LDS-DMA instruction
    first = next((i for i, l in enumerate(ins) if l.startswith('buffer_load_dword ') and l.endswith('lds')), None)
    if first is None:
        print(f"{name[:48]}: no LDS-DMA found"); bad += 1; continue
    # the stage loop proper starts at the first barrier behind it (before that: the chunk prologue, where a full wait
    # for the loads of the kernel prologue is legitimate)
    first = next((i for i in range(first, len(ins)) if ins[i].startswith('s_barrier')), len(ins))
    # (a two-slot ring's own trip-top wait is `s_waitcnt vmcnt(0)` + `s_barrier`: hand-placed, not an alias wait)
    # (and a read that is scalarised right away — v_readfirstlane — is a LIST entry read of a prologue, not an operand)
    hits = [i for i in range(first, len(ins) - 2)
            if ins[i].startswith('s_waitcnt vmcnt(0)') and not ins[i + 1].startswith('s_barrier')
            and any(x.startswith('ds_read_b128') for x in ins[i + 1:i + 3])
            and not any(x.startswith('v_readfirstlane') for x in ins[i + 1:i + 7])]
    print(f"{name[:48]}: {len(hits)} vmcnt(0) in front of ds_read_b128 inside the stage loop")
    bad += len(hits)
sys.exit(1 if bad else 0)
PY
rc=$?
rm -rf "$tmp"
exit $rc
