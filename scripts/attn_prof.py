"""Attention-only workload for rocprofv3 (kernel trace / PMC passes): LM-shaped (D=128, ~790-token docs and
plain causal) and tower-shaped (D=64, T=1500 causal) forward + backward."""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402

dev, bf = "cuda", torch.bfloat16


def docs(B, T, mean_len, seed=0):
    rng = np.random.RandomState(seed)
    out = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = max(1, int(rng.normal(mean_len, mean_len * 0.1)))
            out[b, t:t + n] = d
            t += n
            d += 1
    return torch.from_numpy(out)


def run(B, T, Nh, Nkv, D, doc, iters=3):
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev, requires_grad=True)
    k = torch.randn(B, T, Nkv, D, dtype=bf, device=dev, requires_grad=True)
    v = torch.randn(B, T, Nkv, D, dtype=bf, device=dev, requires_grad=True)
    mask = F.build_packed_mask(doc.to(dev))
    for _ in range(iters):
        o = F.packed_attention(q, k, v, mask)
        torch.autograd.grad(o, [q, k, v], torch.randn_like(o))
    torch.cuda.synchronize()


which = sys.argv[1] if len(sys.argv) > 1 else "all"      # docs | causal | tower | all | long (config-D-like length)
if which == "long":
    run(1, 32768, 32, 32, 128, torch.ones(1, 32768, dtype=torch.int64), iters=2)
if which in ("docs", "all"):
    run(2, 8192, 32, 32, 128, docs(2, 8192, 790))
if which in ("causal", "all"):
    run(2, 8192, 32, 32, 128, torch.ones(2, 8192, dtype=torch.int64))
if which in ("tower", "all"):
    run(20, 1500, 20, 20, 64, torch.ones(20, 1500, dtype=torch.int64))
