"""SwiGLU kernels with transposed second outputs vs the plain kernels + separate transposes, [16384, 11008] bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import _C  # noqa: E402

M, I = 16384, 11008
dev, bf = "cuda", torch.bfloat16
gate, up, d = [torch.randn(M, I, dtype=bf, device=dev) for _ in range(3)]
act, dg, du = [torch.empty(M, I, dtype=bf, device=dev) for _ in range(3)]
act_t = torch.empty(I, M, dtype=bf, device=dev)
dgu_t = torch.empty(2 * I, M, dtype=bf, device=dev)
lib, p, st = _C.lib(), _C.ptr, _C.stream


def t_ms(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


GB = M * I * 2 / 1e9
a = t_ms(lambda: lib.tn_swiglu_fwd(p(gate), p(up), p(act), M * I, 1, st()))
b = t_ms(lambda: F.transpose_2d(act, out=act_t))
c = t_ms(lambda: lib.tn_swiglu_fwd_t(p(gate), p(up), p(act), p(act_t), M, I, st()))
print(f"fwd: plain {a:.3f} ms ({3 * GB / a:.2f} TB/s) + transpose {b:.3f} ms = {a + b:.3f} | fused {c:.3f} ms ({4 * GB / c:.2f} TB/s)")
a = t_ms(lambda: lib.tn_swiglu_bwd(p(d), p(gate), p(up), p(dg), p(du), M * I, 1, st()))
b = t_ms(lambda: (F.transpose_2d(dg, out=dgu_t[:I]), F.transpose_2d(du, out=dgu_t[I:])))
c = t_ms(lambda: lib.tn_swiglu_bwd_t(p(d), p(gate), p(up), p(dg), p(du), p(dgu_t), M, I, st()))
print(f"bwd: plain {a:.3f} ms ({5 * GB / a:.2f} TB/s) + 2 transposes {b:.3f} ms = {a + b:.3f} | fused {c:.3f} ms ({7 * GB / c:.2f} TB/s)")
