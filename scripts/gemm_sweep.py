"""Hand-written MFMA GEMM (csrc/gemm.hip): correctness of every operand mode and a timing sweep over the kernel's
schedule variants against hipBLASLt, on the shapes of the Qwen2-Audio-7B step.  One box, one process, interleaved rounds.

    TN_AMD_LIB=touchnet_amd/_lib/variants/allv/libtouchnet_amd.so python scripts/gemm_sweep.py \
        [--variants 0,100,110,...] [--rounds 3] [--iters 10] [--out gpurun_out/gemm_sweep.json]

(the `allv` library is built with scripts/build_variant.sh allv -DTN_GEMM_ALL_VARIANTS; the product library only holds
the default variant).  variant = 100 * PLACE + 10 * ASYM + ILV, see gemm.hip; --persist 0,1 also times the one-workgroup-per-tile launch.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd.utils import gemm_tuning  # noqa: E402

DEV = "cuda"
T, H, I = 16384, 4096, 11008
# (name, mode, M, N, K): mode fwd = x[M,K] W[N,K]; dgrad = dy[M,K] W[K,N]; wgrad = dy[K,M] x[K,N]
SHAPES = [
    ("fwd   q/o   16384x4096x4096", "fwd", T, H, H),
    ("fwd   gate  16384x11008x4096", "fwd", T, I, H),
    ("fwd   down  16384x4096x11008", "fwd", T, H, I),
    ("dgrad q/o   16384x4096x4096", "dgrad", T, H, H),
    ("dgrad gate  16384x4096x11008", "dgrad", T, H, I),
    ("dgrad down  16384x11008x4096", "dgrad", T, I, H),
    ("wgrad q/o   4096x4096x16384", "wgrad", H, H, T),
    ("wgrad gate  11008x4096x16384", "wgrad", I, H, T),
    ("wgrad down  4096x11008x16384", "wgrad", H, I, T),
]
ALL_VARIANTS = [0, 1, 100, 101, 110, 111, 301, 311, 401, 411, 501, 511, 601, 611]


def rnd(*shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.rand(*shape, device=DEV, generator=g) * 2 - 1).to(torch.bfloat16)


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def set_variant(v, persist=1):
    os.environ["TN_GEMM_VARIANT"] = str(v)
    os.environ["TN_GEMM_PERSIST"] = str(persist)


def operands(mode, M, N, K, seed):
    """-> (a, b, a_kmaj, b_kmaj, fp32 reference thunk, library thunks {name: fn})"""
    if mode == "fwd":
        a, b = rnd(M, K, seed=seed), rnd(N, K, seed=seed + 1)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        return a, b, False, False, (lambda: a.float() @ b.float().t()), {"lib": lambda: torch.mm(a, b.t(), out=out)}
    if mode == "dgrad":
        a, b = rnd(M, K, seed=seed), rnd(K, N, seed=seed + 1)          # dy [M, K], W [K, N]
        bt = b.t().contiguous()
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        return a, b, False, True, (lambda: a.float() @ b.float()), {
            "lib": lambda: torch.mm(a, b, out=out), "lib_pretransposed": lambda: torch.mm(a, bt.t(), out=out)}
    a, b = rnd(K, M, seed=seed), rnd(K, N, seed=seed + 1)              # dy [K, M], x [K, N]
    at, bt = a.t().contiguous(), b.t().contiguous()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    return a, b, True, True, (lambda: a.float().t() @ b.float()), {
        "lib": lambda: torch.mm(a.t(), b, out=out), "lib_pretransposed": lambda: torch.mm(at, bt.t(), out=out)}


def check(name, got, ref):
    scale = float(ref.abs().max())
    err = float((got.float() - ref).abs().max())
    ok = err <= scale * 2 ** -7
    print(f"  check {name:58s} max|err| {err:9.4g} / scale {scale:9.4g}  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def correctness(variants, persists=(1,)):
    """Every mode on ragged shapes (M, N not multiples of 256, several K depths), bias / accumulate / segments."""
    ok = True
    for v, pers in [(v, q) for v in variants for q in persists]:
        set_variant(v, pers)
        v = f"{v}p{pers}"
        for mode in ("fwd", "dgrad", "wgrad"):
            for (M, N, K) in ((256, 256, 64), (520, 264, 192), (1000, 776, 1088), (2048, 1280, 4096),
                              (4360, 4104, 128), (8200, 8192, 64)):   # > 256 tiles: several tiles per workgroup
                if mode == "wgrad":
                    M = (M + 7) // 8 * 8
                a, b, ak, bk, ref, _ = operands(mode, M, N, K, seed=M + N + K)
                got = F.gemm([(a, b)], ak, bk)
                ok &= check(f"v{v} {mode} {M}x{N}x{K}", got, ref())
        # bias + accumulate + transposed copy, forward mode
        a, b, ak, bk, ref, _ = operands("fwd", 776, 520, 320, seed=5)
        bias = rnd(520, seed=9)
        base = rnd(776, 520, seed=10)
        out = base.clone()
        F.gemm([(a, b)], bias=bias, out=out, accumulate=True)
        ok &= check(f"v{v} fwd bias+accumulate", out, ref() + bias.float() + base.float())
        out_t = torch.empty(520, 776, dtype=torch.bfloat16, device=DEV)
        got = F.gemm([(a, b)], out_t=out_t)
        ok &= check(f"v{v} fwd transposed copy", out_t.t(), ref()) and torch.equal(out_t.t(), got)
        # three segments of different depth and pitch: dX = dQ Wq + dK Wk + dV Wv
        M, N = 1032, 520
        segs, refs = [], 0
        for i, K in enumerate((256, 64, 128)):
            big_a = rnd(M, K + 64, seed=20 + i)
            a_s, b_s = big_a[:, :K], rnd(K, N, seed=30 + i)
            segs.append((a_s, b_s))
            refs = refs + a_s.float() @ b_s.float()
        ok &= check(f"v{v} dgrad 3 segments", F.gemm(segs, False, True), refs)
        segs, refs = [], 0
        for i, K in enumerate((128, 192)):
            a_s, b_s = rnd(K, M, seed=40 + i), rnd(K, N, seed=50 + i)
            segs.append((a_s, b_s))
            refs = refs + a_s.float().t() @ b_s.float()
        ok &= check(f"v{v} wgrad 2 segments", F.gemm(segs, True, True), refs)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="")
    ap.add_argument("--check-variants", default="")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default="")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--persist", default="1")
    args = ap.parse_args()
    gemm_tuning.enable()
    variants = [int(v) for v in args.variants.split(",")] if args.variants else ALL_VARIANTS
    cvars = [int(v) for v in args.check_variants.split(",")] if args.check_variants else variants
    persists = [int(v) for v in args.persist.split(",")]
    ok = True
    if not args.no_check:
        ok = correctness(cvars, persists)
        print("CORRECTNESS:", "all ok" if ok else "FAILURES", flush=True)
    res = []
    for name, mode, M, N, K in SHAPES:
        if args.shapes and not any(s in name for s in args.shapes.split(",")):
            continue
        a, b, ak, bk, _, libs = operands(mode, M, N, K, seed=M + N + K)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        fl = 2.0 * M * N * K
        fns = dict(libs)
        for v in variants:
            for q in persists:
                fns[f"own_v{v}p{q}"] = (lambda v=v, q=q: (set_variant(v, q), F.gemm([(a, b)], ak, bk, out=out)))
        for fn in fns.values():
            fn(), fn()
        times = {k: [] for k in fns}
        for _ in range(args.rounds):
            for k, fn in fns.items():
                times[k].append(timeit(fn, args.iters))
        row = {"shape": name, "mode": mode, "M": M, "N": N, "K": K,
               "tf": {k: fl / min(t) / 1e9 for k, t in times.items()},
               "ms": {k: min(t) for k, t in times.items()}}
        res.append(row)
        best = max((k for k in row["tf"] if k.startswith("own")), key=lambda k: row["tf"][k])
        libbest = max((k for k in row["tf"] if k.startswith("lib")), key=lambda k: row["tf"][k])
        print(f"{name:32s} " + " ".join(f"{k}={v:6.0f}" for k, v in row["tf"].items()), flush=True)
        print(f"{'':32s} best own {best} {row['tf'][best]:.0f} TF vs {libbest} {row['tf'][libbest]:.0f} TF "
              f"({row['tf'][best] / row['tf'][libbest]:.3f}x)", flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"correct": ok, "rows": res}, f, indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
