"""Persistent GEMM (one workgroup per CU, fixed tile list) vs one workgroup per tile while ANOTHER kernel holds CUs — the
situation the Trainer switches the launch form for when collectives run beside the compute (bin/train.py; RCCL's
reduce-scatter / all-gather kernels).  The stand-in for the collective is a side-stream device copy in reduce-scatter-sized
pieces (a block's fp32 gradient bucket of the 7B model: 0.8 GB), issued continuously beside a loop of MLP-shaped GEMMs.
Prints ms per GEMM loop for {persistent, per-tile} x {alone, beside the copies}."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import touchnet_amd.functional as F
from touchnet_amd import _C
dev, bf = "cuda", torch.bfloat16
M, H, I = 16384, 4096, 11008
x = torch.randn(M, H, dtype=bf, device=dev); w1 = torch.randn(I, H, dtype=bf, device=dev) * 0.02
dy = torch.randn(M, I, dtype=bf, device=dev)
src = torch.empty(200 * 2 ** 20, dtype=torch.float32, device=dev); dst = torch.empty_like(src)      # 0.8 GB pieces
side = torch.cuda.Stream()

def gemms(n=24):
    for _ in range(n // 3):
        F.gemm([(x, w1)], False, False)            # forward   x W^T
        F.gemm([(dy, w1)], False, True)            # dgrad     dY W
        F.gemm([(dy, x)], True, True)              # wgrad     dY^T x

def run(persist, beside):
    _C.lib().tn_gemm_set_persistent(persist)
    gemms(6); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event()
    if beside:
        with torch.cuda.stream(side):
            for _ in range(40):                    # ~32 GB of copies: longer than the GEMM loop
                dst.copy_(src, non_blocking=True)
    s.record(); gemms(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)

for persist in (1, 0):
    for beside in (False, True):
        ms = min(run(persist, beside) for _ in range(3))
        print(f"persistent={persist} beside_copies={beside}: {ms:.2f} ms per 24 GEMMs ({ms / 24:.3f} ms each)", flush=True)

# ---- a neighbour that HOLDS n CUs for the whole loop (what an RCCL kernel with n channels does)
import ctypes, subprocess
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/hold_cus.so"
subprocess.run(["hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "microbench", "hold_cus.hip"),
                "-o", so], check=True)
hold = ctypes.CDLL(so).hold_cus
hold.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]


def run_held(persist, n):
    _C.lib().tn_gemm_set_persistent(persist)
    gemms(6); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if n:
        assert hold(n, 60000.0, ctypes.c_void_p(side.cuda_stream)) == 0       # 60 ms: longer than the loop
        torch.cuda._sleep(2_000_000)                                           # (let the holders start first)
    s.record(); gemms(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)


for n in (0, 8, 16, 32, 64):
    a, b = (min(run_held(p, n) for _ in range(3)) for p in (1, 0))
    print(f"{n:3d} CUs held: persistent {a:.2f} ms, one workgroup per tile {b:.2f} ms per 24 GEMMs "
          f"(per-tile / persistent = {b / a:.3f})", flush=True)
