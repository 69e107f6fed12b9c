"""Audio tower GEMM shapes (30000 frames, 1280 / 5120 wide, shallow contractions): hand-written kernel vs hipBLASLt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd.utils import gemm_tuning  # noqa: E402

gemm_tuning.enable()
DEV = "cuda"


def t_ms(fn, it=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


R = 30000
for name, N, K in (("q/k/v/o", 1280, 1280), ("fc1", 5120, 1280), ("fc2", 1280, 5120), ("projector", 4096, 1280)):
    x = torch.randn(R if name != "projector" else 15000, K, device=DEV, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=DEV, dtype=torch.bfloat16)
    dy = torch.randn(x.shape[0], N, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * x.shape[0] * N * K
    out = torch.empty(x.shape[0], N, device=DEV, dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    wt = w.t().contiguous()
    a = t_ms(lambda: F.gemm([(x, w)], out=out))
    b = t_ms(lambda: torch.mm(x, w.t(), out=out))
    c = t_ms(lambda: F.gemm([(dy, w)], b_kmaj=True, out=dx))
    d = t_ms(lambda: torch.mm(dy, w, out=dx))
    e = t_ms(lambda: torch.mm(dy, wt.t(), out=dx))
    print(f"{name:10s} [{x.shape[0]}x{N}x{K}] fwd own {fl / a / 1e9:5.0f} lib {fl / b / 1e9:5.0f} TF | dgrad own {fl / c / 1e9:5.0f} "
          f"lib {fl / d / 1e9:5.0f} lib(pre-transposed W) {fl / e / 1e9:5.0f} TF", flush=True)
