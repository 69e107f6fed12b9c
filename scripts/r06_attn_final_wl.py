"""Round 6 PMC workload, final kernels: attention forward + backward (with the rotary gradient in the epilogues at D = 128) on the
headline's two shapes — decoder (1 x 15872 rows of ~790-token documents, 32 heads, D = 128) and audio tower (1 x 30000 frames
of 1500-frame clips, 20 heads, D = 64) — three launches each, default kernel selection."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402

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


for (B, T, Nh, D, mean) in ((1, 15872, 32, 128, 790), (1, 30000, 20, 64, 1500)):
    q, k, v = [torch.randn(B, T, Nh, D, dtype=bf, device=dev).requires_grad_() for _ in range(3)]
    do = torch.randn(B, T, Nh, D, dtype=bf, device=dev)
    doc = (torch.arange(T, device=dev, dtype=torch.int32) // 1500 + 1)[None].contiguous() if mean == 1500 else docs(B, T, mean)
    mask = F.build_packed_mask(doc)
    rope = None
    if D == 128:
        pos = torch.arange(T, device=dev)[None]
        rope = F.rope_tables(pos, F.rope_inv_freq(D, 1e6, device=dev), bf)
    for _ in range(3):
        o = F.packed_attention(q, k, v, mask, rope_grad=rope) if rope is not None else F.packed_attention(q, k, v, mask)
        o.backward(do)
        q.grad = k.grad = v.grad = None
torch.cuda.synchronize()
