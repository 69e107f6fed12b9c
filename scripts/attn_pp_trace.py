"""Per-phase s_memtime trace of the ping-pong attention forward (variant built with -DTN_PP_TRACE).
Prints, for the heaviest causal block, the average cycles of each piece for group A (wave 0) and B (wave 4)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402
from touchnet_amd import _C  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B, T, Nh = 2, 8192, 32
dev, bf = "cuda", torch.bfloat16
q, k, v = [torch.randn(B, T, Nh, D, dtype=bf, device=dev) for _ in range(3)]
mask = F.build_packed_mask(torch.ones(B, T, dtype=torch.int32, device=dev))
with torch.no_grad():
    for _ in range(3):
        F.packed_attention(q, k, v, mask)
torch.cuda.synchronize()
buf = np.zeros((2, 192, 8), dtype=np.uint64)
lib = _C.lib()
lib.tn_debug_pp_trace.argtypes = [ctypes.c_void_p]
assert lib.tn_debug_pp_trace(buf.ctypes.data) == 0
x = buf.astype(np.int64)
names = ["pv", "qk", "->barrier1", "softmax", "store", "issue+advance", "->barrier2", "loop back"]
for g in range(2):
    d = x[g, 8:120]                                   # steady state
    seg = [d[:, i + 1] - d[:, i] for i in range(7)] + [d[1:, 0] - d[:-1, 7]]
    print(f"group {'AB'[g]}: " + "  ".join(f"{n}={np.mean(s):.0f}" for n, s in zip(names, seg)) +
          f"  | per tile {np.mean(d[1:, 0] - d[:-1, 0]):.0f} ticks")
blk = np.zeros((8192, 2), dtype=np.uint64)
lib.tn_debug_pp_blocks.argtypes = [ctypes.c_void_p]
assert lib.tn_debug_pp_blocks(blk.ctypes.data) == 0
nb = 2 * Nh * ((T + 255) // 256)
bb = blk[:nb].astype(np.int64)
t0, t1 = bb[:, 0].min(), bb[:, 1].max()
dur = bb[:, 1] - bb[:, 0]
busy = dur.sum() / 256.0                              # one workgroup per CU at a time
print(f"kernel span {t1 - t0} realtime ticks (100 MHz => {(t1 - t0) / 100:.1f} us); sum of block durations / 256 CUs = "
      f"{busy:.0f} ticks => balance efficiency {busy / (t1 - t0):.3f}; longest block {dur.max()} ticks")
starts = np.sort(bb[:, 0] - t0)
print("block start times (ticks) at ranks 0,255,256,511,1024,2047:", [int(starts[i]) for i in (0, 255, 256, 511, 1024, nb - 1)])
c = x[0, 191]
if c[3] > c[1]:
    print(f"clock probe: {c[2] - c[0]} memtime ticks over {c[3] - c[1]} realtime ticks => shader clock ~ "
          f"{(c[2] - c[0]) / (c[3] - c[1]) * 100:.0f} MHz if realtime is 100 MHz")
print("s_memtime ticks (100 MHz const clock on gfx950? compare the ratio, and the per-tile total with the kernel time)")
