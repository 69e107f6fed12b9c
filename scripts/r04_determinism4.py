"""Reproducibility stress of the attention kernels at the shapes the tests do not repeat: long sequences (ping-pong forward,
fused dK+dV, dQ), GQA, D = 64 tower shapes, bidirectional clips, context-parallel segments — N launches each, bitwise."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import touchnet_amd.functional as F
dev = "cuda"
def docs(B, T, seed, maxlen):
    rng = np.random.RandomState(seed)
    out = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = int(rng.randint(1, maxlen + 1)); out[b, t:min(t + n, T)] = d; t += n; d += 1
    return torch.from_numpy(out).to(dev)
def case(name, B, T, Nh, Nkv, D, maxdoc, reps, bidir=False):
    g = torch.Generator().manual_seed(T + D)
    q = torch.randn(B, T, Nh, D, generator=g).bfloat16().to(dev)
    k = torch.randn(B, T, Nkv, D, generator=g).bfloat16().to(dev)
    v = torch.randn(B, T, Nkv, D, generator=g).bfloat16().to(dev)
    do = torch.randn(B, T, Nh, D, generator=g).bfloat16().to(dev)
    mask = F.build_packed_mask(docs(B, T, 3, maxdoc))
    fn = F.bidirectional_attention if bidir else F.packed_attention
    ref, bad = None, 0
    for it in range(reps):
        qd, kd, vd = [t.clone().requires_grad_(True) for t in (q, k, v)]
        o = fn(qd, kd, vd, mask); o.backward(do)
        cur = [o.detach(), qd.grad, kd.grad, vd.grad]
        if ref is None: ref = cur; continue
        if not all(torch.equal(a, b) for a, b in zip(cur, ref)):
            bad += 1
    print(f"{name}: {'reproducible' if not bad else 'NOT REPRODUCIBLE in ' + str(bad)} over {reps} launches", flush=True)
case("T=32768 causal D128", 1, 32768, 8, 8, 128, 32768, 12)
case("T=32768 docs~9000 D128 GQA 28/4", 1, 32768, 28, 4, 128, 9000, 10)
case("packed 2x8192 docs~790 D128", 2, 8192, 32, 32, 128, 790, 60)
case("packed 2x8192 docs~100 D128", 2, 8192, 32, 32, 128, 100, 60)
case("packed 4x8192 docs~400 D64 GQA 28/4", 4, 8192, 28, 4, 64, 400, 60)
case("tower 20x1500 D64", 20, 1500, 20, 20, 64, 1500, 60)
case("speech clips 16x1500 D64 bidirectional", 16, 1500, 20, 20, 64, 1500, 60, bidir=True)
case("bidirectional docs~300 D128", 2, 2048, 8, 4, 128, 300, 100, bidir=True)
case("tiny docs 4x512 D128", 4, 512, 20, 20, 128, 7, 300)
case("ragged T=1000 D64", 3, 1000, 4, 2, 64, 90, 300)
