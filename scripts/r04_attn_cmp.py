"""Fused dK+dV pass against the two-launch scheme on the same inputs (same library, TN_ATTN_BWD_KV read per call):
max |difference| of dK / dV relative to the tensor's scale, for a few mask shapes."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import touchnet_amd.functional as F
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
def docs(B, T, mean):
    rng = np.random.RandomState(0); out = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = max(1, int(rng.normal(mean, mean * 0.1))); out[b, t:t + n] = d; t += n; d += 1
    return torch.from_numpy(out).to(dev)
for (B, T, Nh, Nkv, D, mean) in ((1, 4096, 4, 4, 128, 0), (1, 4096, 8, 2, 128, 0), (2, 2048, 4, 4, 128, 300), (1, 256, 4, 2, 128, 100), (1, 512, 2, 2, 128, 0)):
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev); k, v = [torch.randn(B, T, Nkv, D, dtype=bf, device=dev) for _ in range(2)]
    doc = torch.ones(B, T, dtype=torch.int32, device=dev) if mean == 0 else docs(B, T, mean)
    mask = F.build_packed_mask(doc)
    res = {}
    for mode in ("split", "fused"):
        os.environ["TN_ATTN_BWD_KV"] = mode
        qg, kg, vg = [x.clone().requires_grad_() for x in (q, k, v)]
        out = F.packed_attention(qg, kg, vg, mask); do = torch.ones_like(out) * 0.5 + q * 0.25
        res[mode] = torch.autograd.grad(out, (qg, kg, vg), do)
    torch.cuda.synchronize()
    for name, a, b_ in zip(("dQ", "dK", "dV"), res["split"], res["fused"]):
        d = (a.float() - b_.float()).abs(); sc = float(a.float().abs().max())
        bad = (d > 0.02 * sc).nonzero()
        first = tuple(int(x) for x in bad[0]) if len(bad) else None
        print(f"T{T} Nh{Nh}/{Nkv} docs~{mean or 'causal'} {name}: max diff {float(d.max()):.4g} (scale {sc:.3g}), {len(bad)} elements > 2 % of scale, first {first}", flush=True)
    if len(bad):
        kvrows = torch.unique(bad[:, 1]); print("   bad kv rows:", kvrows[:40].tolist(), "... n =", len(kvrows), " bad d:", torch.unique(bad[:, 3])[:16].tolist(), flush=True)
