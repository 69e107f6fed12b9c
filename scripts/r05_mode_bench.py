"""Round 5 probe: the hand-written GEMM's operand modes at the SAME product — is the contraction-major B operand (dgrad:
W read as stored, ds_read_b64_tr_b16) slower than the row mode on a pre-transposed W, and by how much?  (If it were, one
transposed weight copy per step — 14 GB of traffic — could buy it back.)  Also the weight-gradient mode against row mode
on pre-transposed operands.
    python scripts/r05_mode_bench.py [rounds [iters]]      (short runs for the PMC passes of scripts/r05_gemm16_pmc.sh)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402


def timeit(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    r = lambda *s: ((torch.rand(*s, device=dev) * 2 - 1)).to(torch.bfloat16)
    M, H, I = 15872, 4096, 11008
    res = {}
    cases = []
    dy, w = r(M, I), r(I, H)                      # dX = dY W : [M, I] x [I, H]
    wt = w.t().contiguous()                       # [H, I]
    cases.append(("dgrad MLP  B contraction-major (as stored)", lambda: F.gemm([(dy, w)], b_kmaj=True), 2.0 * M * I * H))
    cases.append(("dgrad MLP  row mode on pre-transposed W", lambda: F.gemm([(dy, wt)]), 2.0 * M * I * H))
    dq, wq = r(M, H), r(H, H)
    wqt = wq.t().contiguous()
    cases.append(("dgrad attn B contraction-major", lambda: F.gemm([(dq, wq)], b_kmaj=True), 2.0 * M * H * H))
    cases.append(("dgrad attn row mode pre-transposed", lambda: F.gemm([(dq, wqt)]), 2.0 * M * H * H))
    x = r(M, H)
    dyt, xt = dy.t().contiguous(), x.t().contiguous()
    cases.append(("wgrad MLP  both contraction-major (as stored)", lambda: F.gemm([(dy, x)], True, True), 2.0 * M * I * H))
    cases.append(("wgrad MLP  row mode on pre-transposed dY, x", lambda: F.gemm([(dyt, xt)]), 2.0 * M * I * H))
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    for rd in range(rounds):
        for name, fn, fl in cases:
            res.setdefault(name, []).append(timeit(fn, iters))
    for name, fn, fl in cases:
        ts = sorted(res[name])
        print(f"{name:52s} median {ts[len(ts) // 2]:7.3f} ms  min {ts[0]:7.3f}  {fl / ts[len(ts) // 2] / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
