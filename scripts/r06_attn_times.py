"""Round 6: attention forward / backward times on the headline's two shapes with whichever library is loaded (TN_AMD_LIB selects a
variant) — for same-box A/B runs of two builds in alternating processes.  usage: r06_attn_times.py [tag]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402

dev, bf = "cuda", torch.bfloat16
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("TN_AMD_LIB", "default")


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


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


out = []
for name, (B, T, Nh, D, mean) in (("decoder", (1, 15872, 32, 128, 790)), ("tower", (1, 30000, 20, 64, 1500))):
    q, k, v = [torch.randn(B, T, Nh, D, dtype=bf, device=dev).requires_grad_() for _ in range(3)]
    do = torch.randn(B, T, Nh, D, dtype=bf, device=dev)
    doc = (torch.arange(T, device=dev, dtype=torch.int32) // 1500 + 1)[None].contiguous() if mean == 1500 else docs(B, T, mean)
    mask = F.build_packed_mask(doc)
    with torch.no_grad():
        tf = timed(lambda: F.packed_attention(q, k, v, mask))
    o = F.packed_attention(q, k, v, mask)
    tb = timed(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
    out.append(f"{name}: fwd {tf:7.1f} us  bwd {tb:7.1f} us")
print(f"[{tag}] " + " | ".join(out), flush=True)
