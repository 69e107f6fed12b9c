"""Does an HBM-bound stream (AdamW-shaped: read 14 B + write 14 B per element) hide under MFMA-bound GEMMs on MI355X?
Times a GEMM loop alone, the streaming kernel alone, and both on two streams.  (Decides whether running the optimizer of
step i under the forward of step i+1 can pay: DESIGN.md 8.)"""
import torch

dev = "cuda"
M, K, N = 16384, 4096, 11008
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
n = 1_500_000_000
p = torch.zeros(n, device=dev)
m = torch.zeros(n, device=dev)
v = torch.zeros(n, device=dev)
g = torch.zeros(n, device=dev, dtype=torch.bfloat16)


def gemms(k=40):
    for _ in range(k):
        torch.nn.functional.linear(a, w)


def stream_pass():          # 3 fp32 read-modify-write streams + a bf16 read: the optimizer's traffic shape
    torch._foreach_mul_([p, m, v], 0.999)
    p.add_(g, alpha=1e-3)


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e)


side = torch.cuda.Stream()


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        stream_pass()
    gemms()
    torch.cuda.current_stream().wait_stream(side)


for f in (gemms, stream_pass, both):
    f()
for name, f in (("gemm loop alone", gemms), ("stream alone", stream_pass), ("both, two streams", both),
                ("gemm loop alone", gemms), ("both, two streams", both)):
    print(f"{name:20s} {timed(f):8.2f} ms", flush=True)
