"""Round 6 PMC workload: the attention FORWARD on the headline's two shapes — decoder (1 x 15872 rows of ~790-token
documents, 32 heads, D = 128) and audio tower (1 x 30000 frames of 1500-frame clips, 20 heads, D = 64) — under schedule 0
(attn_fwd.hip) and schedule 2 (attn_fwd_stream.hip), three launches each.  Usage: r06_attn_fwd_wl.py [0|2|both]"""
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


which = sys.argv[1] if len(sys.argv) > 1 else "both"
for (B, T, Nh, D, mean) in ((1, 15872, 32, 128, 790), (1, 30000, 20, 64, 1500)):
    q, k, v = [torch.randn(B, T, Nh, D, dtype=bf, device=dev) for _ in range(3)]
    if mean == 1500:
        doc = (torch.arange(T, device=dev, dtype=torch.int32) // 1500 + 1)[None].contiguous()
    else:
        doc = docs(B, T, mean)
    mask = F.build_packed_mask(doc)
    for sched in ([0, 2] if which == "both" else [int(which)]):
        _C.lib().tn_attn_set_fwd_schedule(sched)
        for _ in range(3):
            L.attn_fwd(q, k, v, mask.doc, mask.meta, D ** -0.5)
        torch.cuda.synchronize()
_C.lib().tn_attn_set_fwd_schedule(-2)
