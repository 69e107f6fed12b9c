"""Fixed cost per workgroup of the fused dK+dV kernel: the backward of one packed shape, repeated (run under rocprofv3 --stats
with the default library and with the -DTN_FKV_ABL=64 variant, which skips the trips)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import touchnet_amd.functional as F
maxdoc = int(sys.argv[1]) if len(sys.argv) > 1 else 790
B, T, Nh, D = 2, 8192, 32, 128
rng = np.random.RandomState(1)
doc = np.zeros((B, T), dtype=np.int64)
for b in range(B):
    t, d = 0, 1
    while t < T:
        n = int(rng.randint(max(1, maxdoc // 2), maxdoc + 1)) if maxdoc < T else T
        doc[b, t:t + n] = d; t += n; d += 1
mask = F.build_packed_mask(torch.from_numpy(doc).cuda())
g = torch.Generator().manual_seed(0)
q, k, v, do = [torch.randn(B, T, Nh, D, generator=g).bfloat16().cuda() for _ in range(4)]
for _ in range(12):
    qd, kd, vd = [t.clone().requires_grad_(True) for t in (q, k, v)]
    F.packed_attention(qd, kd, vd, mask).backward(do)
torch.cuda.synchronize()
