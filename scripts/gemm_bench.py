"""Hand-written MFMA GEMM (csrc/gemm.hip) vs the library GEMM (torch.mm -> hipBLASLt) on the shapes of the
Qwen2-Audio-7B step, same box, same random operands, interleaved rounds (HIP events on the current stream).

    python scripts/gemm_bench.py [--rounds 5] [--iters 20] [--out gpurun_out/gemm_bench.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd.utils import gemm_tuning  # noqa: E402

SHAPES = [  # (name, M, N, K)
    ("attn_o/q  16384x4096x4096", 16384, 4096, 4096),
    ("qkv       16384x12288x4096", 16384, 12288, 4096),
    ("gate_up   16384x22016x4096", 16384, 22016, 4096),
    ("down      16384x4096x11008", 16384, 4096, 11008),
    ("wgrad_gu  22016x4096x16384", 22016, 4096, 16384),
    ("wgrad_dn  4096x11008x16384", 4096, 11008, 16384),
    ("square    4096^3", 4096, 4096, 4096),
    ("square    8192^3", 8192, 8192, 8192),
]


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    ap.add_argument("--shapes", default="")
    args = ap.parse_args()
    gemm_tuning.enable()           # the replayed hipBLASLt picks the product uses
    dev = "cuda"
    res = []
    for name, M, N, K in SHAPES:
        if args.shapes and not any(s in name for s in args.shapes.split(",")):
            continue
        g = torch.Generator(device=dev).manual_seed(M + N + K)
        a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        b = (torch.rand(N, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        lib = lambda: torch.mm(a, b.t(), out=out)
        own = lambda: F.gemm_tn(a, b, out=out)
        own()
        got = out.clone()
        lib()
        err = float((got.float() - out.float()).abs().max())
        scale = float(out.float().abs().max())
        for _ in range(3):
            lib(), own()
        t_lib, t_own = [], []
        for _ in range(args.rounds):
            t_lib.append(timeit(lib, args.iters))
            t_own.append(timeit(own, args.iters))
        fl = 2.0 * M * N * K
        r = {"shape": name, "M": M, "N": N, "K": K,
             "hipblaslt_ms": min(t_lib), "own_ms": min(t_own),
             "hipblaslt_tf": fl / min(t_lib) / 1e9, "own_tf": fl / min(t_own) / 1e9,
             "own_tf_median": fl / sorted(t_own)[len(t_own) // 2] / 1e9,
             "max_abs_diff_vs_lib": err, "out_scale": scale}
        res.append(r)
        print(f"{name:32s} hipBLASLt {r['hipblaslt_ms']:.3f} ms {r['hipblaslt_tf']:7.1f} TF | own {r['own_ms']:.3f} ms "
              f"{r['own_tf']:7.1f} TF ({r['own_tf'] / r['hipblaslt_tf']:.2f}x)  diff {err:.3g}/{scale:.3g}", flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
