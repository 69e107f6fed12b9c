"""Ablation timing of the D=128 attention forward (guide §5.4: ablate before optimising): full kernel vs the
kernel with one piece removed.  The drop in time when a piece is removed = what that piece costs in situ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import _C  # noqa: E402

dev, bf = "cuda", torch.bfloat16
B, T, Nh, D = 2, 8192, 32, 128
q, k, v = [torch.randn(B, T, Nh, D, dtype=bf, device=dev) for _ in range(3)]
o = torch.empty_like(q)
lse = torch.empty(B, Nh, T, dtype=torch.float32, device=dev)
names = {0: "full", 1: "no staging (loads+LDS stores)", 2: "no softmax VALU", 3: "1/16 of P.V MFMAs",
         4: "1/8 of QK^T MFMAs", 5: "no barrier", 6: "full, 8 waves x 32 rows (256-row block)"}
import numpy as np
def _docs(mean):
    rng = np.random.RandomState(0); out = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = max(1, int(rng.normal(mean, mean * 0.1))); out[b, t:t + n] = d; t += n; d += 1
    return torch.from_numpy(out).to(dev)
for label, doc in (("causal", torch.ones(B, T, dtype=torch.int32, device=dev)), ("docs~790", _docs(790))):
    mask = F.build_packed_mask(doc)
    for abl in range(7):
        def run():
            _C.check(_C.lib().tn_attn_fwd_ablate(_C.ptr(q), _C.ptr(k), _C.ptr(v), _C.ptr(o), _C.ptr(lse),
                                                 _C.ptr(mask.doc), _C.ptr(mask.meta), B, T, Nh, Nh, D ** -0.5, abl,
                                                 _C.stream()), "ablate")
        for _ in range(3):
            run()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        print(f"{label:8s} abl={abl} {names[abl]:34s} {s.elapsed_time(e) / 10:.3f} ms", flush=True)
