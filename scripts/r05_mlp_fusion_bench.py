"""Round 5: the MLP's fused launches against the launches they replace, same box, interleaved rounds (HIP events):
    fwd   gate+up with the SwiGLU epilogue          vs  2 products + tn_swiglu_fwd
    bwd   d(act) product with the SwiGLU epilogue   vs  1 product + tn_swiglu_bwd
    wgrad the three weight gradients, grouped       vs  3 launches
at the Qwen2-Audio-7B shapes (M = 16384 tokens, hidden 4096, intermediate 11008).
    python scripts/r05_mlp_fusion_bench.py [--rounds 5] [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import library as L  # noqa: E402


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tokens", type=int, default=16384)
    a = ap.parse_args()
    M, H, I = a.tokens, 4096, 11008
    dev = "cuda"
    r = lambda *s, sc=1.0: ((torch.rand(*s, device=dev) * 2 - 1) * sc).to(torch.bfloat16)
    x, wg, wu, wd = r(M, H), r(I, H, sc=0.03), r(I, H, sc=0.03), r(H, I, sc=0.02)
    dy = r(M, H)
    gate, up, act = F.gemm_swiglu_fwd(x, wg, wu)
    dgate, dup = F.gemm_swiglu_bwd(dy, wd, gate, up)

    def fwd_old():
        g_, u_ = F.gemm([(x, wg)]), F.gemm([(x, wu)])
        return L.swiglu_fwd(g_, u_)

    def bwd_old():
        return L.swiglu_bwd(F.gemm([(dy, wd)], b_kmaj=True), gate, up)

    def wg_old():
        return F.gemm([(dgate, x)], True, True), F.gemm([(dup, x)], True, True), F.gemm([(dy, act)], True, True)

    cases = [
        ("fwd  gate+up+SwiGLU epilogue (1 launch)", lambda: F.gemm_swiglu_fwd(x, wg, wu), 4.0 * M * H * I),
        ("fwd  2 products + swiglu kernel", fwd_old, 4.0 * M * H * I),
        ("bwd  d(act)+SwiGLU-bwd epilogue (1 launch)", lambda: F.gemm_swiglu_bwd(dy, wd, gate, up), 2.0 * M * H * I),
        ("bwd  1 product + swiglu_bwd kernel", bwd_old, 2.0 * M * H * I),
        ("wgrad 3 weight gradients grouped (+ split-K remainder)", lambda: F.gemm_grouped_wgrad([(dgate, x), (dup, x), (dy, act)]), 6.0 * M * H * I),
        ("wgrad 3 launches", wg_old, 6.0 * M * H * I),
    ]
    best = {}
    for rd in range(a.rounds):
        for name, fn, fl in cases:
            t = timeit(fn, a.iters)
            best.setdefault(name, []).append(t)
    for name, fn, fl in cases:
        ts = sorted(best[name])
        med = ts[len(ts) // 2]
        print(f"{name:58s} median {med:7.3f} ms  min {ts[0]:7.3f}  max {ts[-1]:7.3f}   {fl / med / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
