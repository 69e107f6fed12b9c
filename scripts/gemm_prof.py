"""GEMM-only workload for rocprofv3 (kernel trace / PMC passes): the hand-written kernel (forward mode and the
contraction-major weight-gradient mode) and the library GEMM on decoder-block shapes.   python scripts/gemm_prof.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd.utils import gemm_tuning  # noqa: E402

gemm_tuning.enable()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
for M, N, K in [(16384, 4096, 4096), (16384, 4096, 11008)]:
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(iters):
        F.gemm([(a, b)], out=out)
        torch.mm(a, b.t(), out=out)
# weight gradient: dy [tokens, N]^T x [tokens, K]
M, N, K = 4096, 4096, 16384
dy = (torch.rand(K, M, device=dev) * 2 - 1).to(torch.bfloat16)
x = (torch.rand(K, N, device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(iters):
    F.gemm([(dy, x)], True, True, out=out)
torch.cuda.synchronize()
