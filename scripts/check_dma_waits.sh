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
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -S --cuda-device-only \
  "$root/touchnet_amd/csrc/attn_bwd.hip" -o "$tmp/bwd.s" 2>/dev/null
python3 - "$tmp/bwd.s" <<'PY'
import re, sys
text = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r'^(_ZN2tn18attn_bwd_(?:kv|dq)_kernel\S*):.*?s_endpgm', text, re.S | re.M):
    name, body = m.group(1), m.group(0).split('\n')
    ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((';', '.'))]
    # stage loop = everything after the first LDS-DMA instruction
    first = next((i for i, l in enumerate(ins) if l.startswith('buffer_load') and l.endswith('lds')), None)
    if first is None:
        print(f"{name[:48]}: no LDS-DMA found"); bad += 1; continue
    # the stage loop proper starts at the first barrier behind it (before that: the chunk prologue, where a full wait
    # for the K/V register loads of the kernel prologue is legitimate)
    first = next((i for i in range(first, len(ins)) if ins[i].startswith('s_barrier')), len(ins))
    hits = [i for i in range(first, len(ins) - 2)
            if ins[i].startswith('s_waitcnt vmcnt(0)') and any(x.startswith('ds_read_b128') for x in ins[i + 1:i + 3])]
    print(f"{name[:48]}: {len(hits)} vmcnt(0) in front of ds_read_b128 inside the stage loop")
    bad += len(hits)
sys.exit(1 if bad else 0)
PY
rc=$?
rm -rf "$tmp"
exit $rc
