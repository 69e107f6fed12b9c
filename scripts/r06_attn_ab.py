"""Round 6: A/B of the attention forward schedules inside ONE process (same box, same clocks, interleaved).
schedule 0 = attn_fwd.hip (register-staged K/V, one head per workgroup), 2 = attn_fwd_stream.hip (LDS-DMA ring, several
heads per workgroup).  Prints ms, dense-equivalent TFLOP/s, and the max abs difference of O / LSE2 between the schedules
(the stream kernel runs the same tiles through the same arithmetic: expected 0)."""
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


def flops(doc, Nh, D, bidir=False):
    d = doc.cpu().numpy()
    tot = 0
    for b in range(d.shape[0]):
        _, cnt = np.unique(d[b][d[b] > 0], return_counts=True)
        c = cnt.astype(np.int64)
        tot += int((c * c).sum()) if bidir else int((c * (c + 1) // 2).sum())
    return 4.0 * tot * Nh * D


def timed(fn, n=20, warm=3):
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
    (1, 15872, 32, 32, 128, 790, 0),       # the headline's decoder (pad rows dropped, B x T joined)
    (2, 8192, 32, 32, 128, 790, 300),
    (1, 30000, 20, 20, 64, 1500, 0),       # the headline's audio tower
    (2, 8192, 32, 32, 128, 0, 0),
    (2, 8192, 32, 32, 128, 100, 0),
    (4, 8192, 28, 4, 128, 400, 0),         # Kimi GQA
    (2, 8192, 32, 8, 64, 790, 0),          # Llama-1B GQA
    (1, 4000, 5, 5, 64, 333, 77),          # ragged: T % 64 != 0, Nh % hpw != 0
    (1, 32768, 32, 32, 128, 0, 0),         # long rows: the ping-pong kernel's territory (schedule 1)
    (1, 65536, 28, 4, 128, 30000, 0),
]
hpws = [int(x) for x in os.environ.get("HPWS", "1,2,4,8").split(",")]
for (B, T, Nh, Nkv, D, mean, pad) in cases:
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev)
    k, v = [torch.randn(B, T, Nkv, D, dtype=bf, device=dev) for _ in range(2)]
    doc = torch.ones(B, T, dtype=torch.int32, device=dev) if mean == 0 else docs(B, T, mean)
    if pad:
        doc[:, T - pad:] = 0
    mask = F.build_packed_mask(doc)
    fl = flops(doc, Nh, D)
    scale = float(D) ** -0.5
    res = {}
    line = f"B{B} T{T} Nh{Nh}/{Nkv} D{D} docs~{mean or 'causal'} pad{pad}:"
    for sched, hpw in [(0, 0)] + ([(1, 0)] if T >= 32768 else []) + [(2, h) for h in hpws]:
        _C.lib().tn_attn_set_fwd_schedule(sched)
        o, lse = L.attn_fwd(q, k, v, mask.doc, mask.meta, scale)
        torch.cuda.synchronize()
        res[(sched, hpw)] = (o.float().clone(), lse.clone())
        ms = timed(lambda: L.attn_fwd(q, k, v, mask.doc, mask.meta, scale))
        tag = "base" if sched == 0 else ("pingpong" if sched == 1 else f"stream/{hpw}")
        line += f"  {tag} {ms * 1e3:.0f}us {fl / ms / 1e9:.0f}TF"
    o0, l0 = res[(0, 0)]
    worst = 0.0
    for key, (o, l) in res.items():
        if key == (0, 0):
            continue
        fin = torch.isfinite(l0)
        same_inf = bool((torch.isfinite(l) == fin).all())
        worst = max(worst, float((o - o0).abs().max()), float((l[fin] - l0[fin]).abs().max()) if fin.any() else 0.0)
        if not same_inf:
            worst = float("inf")
    print(line + f"  | max|diff| vs base {worst:.3g}", flush=True)
_C.lib().tn_attn_set_fwd_schedule(-2)
