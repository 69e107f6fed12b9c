"""hipBLASLt layout experiments behind functional._LinearGroup: (1) input-gradient GEMM dX = dY W with W [N,K]
as stored (contraction = W's slow dim, "NN") vs with W transposed to [K,N] (forward layout); (2) audio-tower
weight-gradient shapes NT vs TN."""
import torch

dev, bf = "cuda", torch.bfloat16


def bench(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print("== dgrad: dX[M,K] = dY[M,N] @ W[N,K]")
for (M, N, K) in ((16384, 4096, 4096), (16384, 11008, 4096), (16384, 4096, 11008), (30000, 1280, 1280),
                  (30000, 5120, 1280), (30000, 1280, 5120)):
    dY = torch.randn(M, N, dtype=bf, device=dev)
    W = torch.randn(N, K, dtype=bf, device=dev)
    Wt = W.t().contiguous()
    fl = 2.0 * M * N * K
    t_nn = bench(lambda: torch.mm(dY, W))
    t_tn = bench(lambda: torch.mm(dY, Wt.t()))
    print(f"M={M} N={N} K={K}: NN {t_nn:.3f} ms {fl/t_nn/1e9:.0f} TF | W^T (fwd layout) {t_tn:.3f} ms {fl/t_tn/1e9:.0f} TF",
          flush=True)

print("== wgrad: dW[N,K] = dY[M,N]^T X[M,K]")
for (M, N, K) in ((30000, 1280, 1280), (30000, 3840, 1280), (30000, 5120, 1280), (30000, 1280, 5120),
                  (16384, 4096, 11008)):
    dY = torch.randn(M, N, dtype=bf, device=dev)
    X = torch.randn(M, K, dtype=bf, device=dev)
    dYt, Xt = dY.t().contiguous(), X.t().contiguous()
    fl = 2.0 * M * N * K
    t_nt = bench(lambda: torch.mm(dY.t(), X))
    t_tn = bench(lambda: torch.mm(dYt, Xt.t()))
    print(f"M={M} N={N} K={K}: NT {t_nt:.3f} ms {fl/t_nt/1e9:.0f} TF | TN {t_tn:.3f} ms {fl/t_tn/1e9:.0f} TF | "
          f"transposes @5TB/s ~{(M*N+M*K)*4/5e12*1e3:.3f} ms", flush=True)
