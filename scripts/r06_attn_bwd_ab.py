"""Round 6: A/B of the attention BACKWARD variants inside ONE process (same box, interleaved): dQ kernel 0 = attn_bwd.hip,
1 = attn_bwd_dq_stream.hip.  Times the whole backward call (dQ + dK/dV launches; the dK/dV kernel is the same in both arms,
so the difference is the dQ kernel's) and compares dQ / dK / dV / delta between the arms."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import _C  # noqa: E402
from touchnet_amd import library as L  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def docs(B, T, mean, seed=0):
    rng = np.random.RandomState(seed)
    out = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = max(1, int(rng.normal(mean, mean * 0.1)))
            out[b, t:t + n] = d
            t += n
            d += 1
    return torch.from_numpy(out).to(dev)


def flops(doc, Nh, D):
    d = doc.cpu().numpy()
    tot = 0
    for b in range(d.shape[0]):
        _, cnt = np.unique(d[b][d[b] > 0], return_counts=True)
        c = cnt.astype(np.int64)
        tot += int((c * (c + 1) // 2).sum())
    return 4.0 * tot * Nh * D


def timed(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


cases = [  # B, T, Nh, Nkv, D, mean doc length (0 = one document per row), pad tail
    (1, 15872, 32, 32, 128, 790, 0),       # the headline's decoder
    (1, 30000, 20, 20, 64, 1500, 0),       # the headline's audio tower
    (2, 8192, 32, 32, 128, 0, 0),
    (2, 8192, 32, 32, 128, 100, 300),
    (4, 8192, 28, 4, 128, 400, 0),         # Kimi GQA
    (2, 8192, 32, 8, 64, 790, 0),          # Llama-1B GQA
    (1, 4000, 5, 5, 64, 333, 77),          # ragged
    (1, 32768, 32, 32, 128, 0, 0),
]
for (B, T, Nh, Nkv, D, mean, pad) in cases:
    g = torch.Generator(device=dev).manual_seed(T + D)
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev, generator=g)
    k, v = [torch.randn(B, T, Nkv, D, dtype=bf, device=dev, generator=g) for _ in range(2)]
    do = torch.randn(B, T, Nh, D, dtype=bf, device=dev, generator=g)
    doc = torch.ones(B, T, dtype=torch.int32, device=dev) if mean == 0 else docs(B, T, mean)
    if mean == 1500:
        doc = (torch.arange(T, device=dev, dtype=torch.int32) // 1500 + 1)[None].contiguous()
    if pad:
        doc[:, T - pad:] = 0
    mask = F.build_packed_mask(doc)
    scale = float(D) ** -0.5
    o, lse = L.attn_fwd(q, k, v, mask.doc, mask.meta, scale)
    fl = 2.5 * flops(doc, Nh, D)
    res = {}
    line = f"B{B} T{T} Nh{Nh}/{Nkv} D{D} docs~{mean or 'causal'} pad{pad}:"
    for mode in (0, 1):
        _C.lib().tn_attn_set_bwd_dq(mode)
        out = L.attn_bwd(q, k, v, o, do, lse, mask.doc, mask.meta, scale)
        torch.cuda.synchronize()
        res[mode] = [t.float().clone() for t in out]
        ms = timed(lambda: L.attn_bwd(q, k, v, o, do, lse, mask.doc, mask.meta, scale))
        line += f"  dq={'old' if mode == 0 else 'stream'} {ms * 1e3:.0f}us {fl / ms / 1e9:.0f}TF"
    worst = max(float((a - b_).abs().max()) for a, b_ in zip(res[0], res[1]))
    scale_ref = max(float(a.abs().max()) for a in res[0])
    print(line + f"  | max|diff| {worst:.3g} (max|grad| {scale_ref:.3g})", flush=True)
_C.lib().tn_attn_set_bwd_dq(-1)
