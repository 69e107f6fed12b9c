"""In-kernel cycle count (s_memtime) of the 4-wave GEMM built with -DTN_GEMM_ABL4=<mask | 32>: cycles per MFMA per wave."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F
M, N, K = 16384, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    F.gemm([(a, b)], out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); F.gemm([(a, b)], out=out); e1.record(); torch.cuda.synchronize()
cyc = out.view(-1).view(torch.int64)[:1024].cpu().double()
mfma = (M // 256) * (N // 256) / 256 * (K // 64) * 64
print(f"K={K} wall {e0.elapsed_time(e1)*1e3:.0f} us  wave cycles min {cyc.min():.0f} mean {cyc.mean():.0f} max {cyc.max():.0f}  "
      f"-> {cyc.mean()/mfma:.1f} cycles/MFMA  clock {cyc.mean()/e0.elapsed_time(e1)/1e3:.2f} GHz (wave life / kernel time)")
