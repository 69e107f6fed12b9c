"""Forward-attention schedule comparison: run once per TN_ATTN_FWD_SCHEDULE value (0 = baseline 4-wave blocks,
1 = ping-pong 8-wave blocks).  Prints ms and dense-equivalent TFLOP/s per case."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def docs(B, T, mean):
    rng = np.random.RandomState(0)
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
        tot += int((cnt.astype(np.int64) * (cnt + 1) // 2).sum())
    return 4.0 * tot * Nh * D


sched = os.environ.get("TN_ATTN_FWD_SCHEDULE", "default")
for (B, T, Nh, Nkv, D, mean) in ((2, 8192, 32, 32, 128, 0), (2, 8192, 32, 32, 128, 790), (2, 8192, 32, 32, 128, 100),
                                 (2, 8192, 32, 32, 64, 0), (4, 8192, 28, 4, 64, 400), (2, 8192, 32, 8, 64, 790)):
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev)
    k, v = [torch.randn(B, T, Nkv, D, dtype=bf, device=dev) for _ in range(2)]
    doc = torch.ones(B, T, dtype=torch.int32, device=dev) if mean == 0 else docs(B, T, mean)
    mask = F.build_packed_mask(doc)
    with torch.no_grad():
        for _ in range(3):
            F.packed_attention(q, k, v, mask)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            F.packed_attention(q, k, v, mask)
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    # backward (delta + dK/dV + dQ kernels)
    qg, kg, vg = [x.clone().requires_grad_() for x in (q, k, v)]
    out = F.packed_attention(qg, kg, vg, mask)
    do = torch.randn_like(out)
    for _ in range(2):
        torch.autograd.grad(out, (qg, kg, vg), do, retain_graph=True)
    s.record()
    for _ in range(10):
        torch.autograd.grad(out, (qg, kg, vg), do, retain_graph=True)
    e.record()
    torch.cuda.synchronize()
    msb = s.elapsed_time(e) / 10
    fl = flops(doc, Nh, D)
    print(f"sched={sched} B{B} T{T} Nh{Nh}/{Nkv} D{D} docs~{mean or 'causal'}: fwd {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s"
          f" | bwd {msb:.3f} ms {2.5 * fl / msb / 1e9:.1f} TFLOP/s", flush=True)
