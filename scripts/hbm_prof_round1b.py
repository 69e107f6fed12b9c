"""The HBM-bound kernels added late in round 1, a few launches each at Qwen2-Audio-7B shapes, for
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes).  Parsed by scripts/pmc_table.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import _C  # noqa: E402

dev, bf = "cuda", torch.bfloat16
N, H, I = 16384, 4096, 11008
lib, p, st = _C.lib(), _C.ptr, _C.stream
x = torch.randn(N, I, dtype=bf, device=dev)
xt = torch.empty(I, N, dtype=bf, device=dev)
gate, up, d = [torch.randn(N, I, dtype=bf, device=dev) for _ in range(3)]
act, dg, du = [torch.empty(N, I, dtype=bf, device=dev) for _ in range(3)]
dgu_t = torch.empty(2 * I, N, dtype=bf, device=dev)
tower = torch.randn(30000, 1280, dtype=bf, device=dev)
pcm = torch.randint(-30000, 30000, (20 * 480000,), dtype=torch.int16, device=dev)
for _ in range(3):
    F.transpose_2d(x, out=xt)
    lib.tn_swiglu_fwd_t(p(gate), p(up), p(act), p(xt), N, I, st())
    lib.tn_swiglu_bwd_t(p(d), p(gate), p(up), p(dg), p(du), p(dgu_t), N, I, st())
    F.column_sum(tower)
    F.pcm16_to_float(pcm)
torch.cuda.synchronize()
