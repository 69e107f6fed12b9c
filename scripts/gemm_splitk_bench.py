"""Split-K of the hand-written GEMM on the step's few-tile / deep-contraction products (audio tower weight gradients) against
the unsplit kernel and hipBLASLt:  python scripts/gemm_splitk_bench.py"""
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


rows = 30000
for name, M, N in (("tower q/k/v/o wgrad", 1280, 1280), ("tower qkv stacked wgrad", 3840, 1280),
                   ("tower fc1 wgrad", 5120, 1280), ("tower fc2 wgrad", 1280, 5120), ("projector wgrad", 4096, 1280)):
    K = rows if "projector" not in name else 15000
    dy = torch.randn(K, M, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(K, N, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    F.SPLIT_K = True
    parts = F.split_k(M, N, K, True, True)
    a = t_ms(lambda: F.gemm([(dy, x)], True, True))
    F.SPLIT_K = False
    b = t_ms(lambda: F.gemm([(dy, x)], True, True))
    F.SPLIT_K = True
    c = t_ms(lambda: torch.mm(dy.t(), x))
    print(f"{name:26s} [{M}x{N}] K={K}: split-k x{parts} {a * 1e3:7.1f} us {fl / a / 1e9:6.0f} TF | unsplit {b * 1e3:7.1f} us "
          f"{fl / b / 1e9:6.0f} TF | hipBLASLt {c * 1e3:7.1f} us {fl / c / 1e9:6.0f} TF", flush=True)
