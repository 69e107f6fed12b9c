"""Does hipBLASLt run the weight-gradient GEMM faster when both operands are pre-transposed (contraction dim
contiguous, the layout of the forward GEMM)?  dW[N,K] = dY[M,N]^T X[M,K], M = tokens."""
import torch

dev, bf = "cuda", torch.bfloat16
M = 16384


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


for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (22016, 4096), (12288, 4096)):
    dY = torch.randn(M, N, dtype=bf, device=dev)
    X = torch.randn(M, K, dtype=bf, device=dev)
    dYt, Xt = dY.t().contiguous(), X.t().contiguous()
    fl = 2.0 * M * N * K
    t_nt = bench(lambda: torch.mm(dY.t(), X))                 # what autograd does (NT)
    t_tn = bench(lambda: torch.mm(dYt, Xt.t()))               # both contraction-contiguous (TN, like fwd)
    t_a = bench(lambda: torch.mm(dYt, X))                     # only dY transposed
    t_b = bench(lambda: torch.mm(dY.t(), Xt.t()))             # only X transposed
    t_tr = bench(lambda: dY.t().contiguous()) + bench(lambda: X.t().contiguous())
    print(f"dW[{N},{K}] over M={M}: NT {t_nt:.3f} ms {fl/t_nt/1e9:.0f} TF | TN {t_tn:.3f} ms {fl/t_tn/1e9:.0f} TF | "
          f"dYt only {t_a:.3f} | Xt only {t_b:.3f} | torch transposes cost {t_tr:.3f} ms", flush=True)
