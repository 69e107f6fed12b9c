"""HBM-bound kernels at the Qwen2-Audio-7B shapes, a few launches each, for `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` passes (separate passes, as MI355X_MICROARCH.md prescribes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd.utils.optimizer import FusedAdamW  # noqa: E402

dev, bf = "cuda", torch.bfloat16
N, H, I, V = 16384, 4096, 11008, 156032
x = torch.randn(N, H, dtype=bf, device=dev, requires_grad=True)
r = torch.randn(N, H, dtype=bf, device=dev)
w = torch.ones(H, dtype=bf, device=dev, requires_grad=True)
for _ in range(3):
    y, h = F.rms_norm(x, w, 1e-5, residual=r)
    torch.autograd.grad(y, [x, w], torch.randn_like(y))
g, u = [torch.randn(N, I, dtype=bf, device=dev, requires_grad=True) for _ in range(2)]
for _ in range(3):
    o = F.swiglu(g, u)
    torch.autograd.grad(o, [g, u], torch.randn_like(o))
q = torch.randn(2, 8192, 32, 128, dtype=bf, device=dev)
k = torch.randn(2, 8192, 32, 128, dtype=bf, device=dev)
cos, sin = F.rope_tables(torch.arange(8192, device=dev).repeat(2, 1), F.rope_inv_freq(128, 1e4).to(dev), bf)
for _ in range(3):
    F.apply_rope(q, k, cos, sin)
logits = torch.randn(1, 2048, V, dtype=bf, device=dev, requires_grad=True)
labels = torch.randint(0, V, (1, 2048), device=dev)
sl = torch.full((1, 2048), 7, device=dev)
for _ in range(3):
    loss, _ = F.packed_cross_entropy(logits, labels, sl, 10)
    torch.autograd.grad(loss, logits)
p = torch.nn.Parameter(torch.randn(64 * 1024 * 1024, device=dev).bfloat16())
opt = FusedAdamW([p], lr=1e-3)
for _ in range(3):
    p.grad = torch.randn_like(p)
    opt.step()
torch.cuda.synchronize()
