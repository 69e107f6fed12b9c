"""Forward / backward packed attention at config-D-like lengths (T = 65536 plain causal and two 30000-token documents, the
Kimi 28/4 GQA shape at T = 32768): ms, TFLOP/s on the allowed (query, key) pairs and fraction of the 2.5 PF bf16 peak."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import touchnet_amd.functional as F
dev, bf = "cuda", torch.bfloat16
def flops(doc, Nh, D):
    d = doc.cpu().numpy(); tot = 0
    for b in range(d.shape[0]):
        _, cnt = np.unique(d[b][d[b] > 0], return_counts=True)
        tot += int((cnt.astype(np.int64) * (cnt + 1) // 2).sum())
    return 4.0 * tot * Nh * D
for (B, T, Nh, Nkv, D, docs) in ((1, 65536, 32, 32, 128, None), (1, 65536, 32, 32, 128, (30000, 30000, 5536)), (1, 32768, 28, 4, 128, None)):
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev); k, v = [torch.randn(B, T, Nkv, D, dtype=bf, device=dev) for _ in range(2)]
    doc = torch.ones(B, T, dtype=torch.int32, device=dev)
    if docs:
        o = 0
        for i, n in enumerate(docs): doc[:, o:o + n] = i + 1; o += n
    mask = F.build_packed_mask(doc)
    qg, kg, vg = [x.clone().requires_grad_() for x in (q, k, v)]
    out = F.packed_attention(qg, kg, vg, mask); do = torch.randn_like(out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        F.packed_attention(q, k, v, mask); s.record()
        for _ in range(5): F.packed_attention(q, k, v, mask)
        e.record(); torch.cuda.synchronize(); ms = s.elapsed_time(e) / 5
    torch.autograd.grad(out, (qg, kg, vg), do, retain_graph=True); s.record()
    for _ in range(3): torch.autograd.grad(out, (qg, kg, vg), do, retain_graph=True)
    e.record(); torch.cuda.synchronize(); msb = s.elapsed_time(e) / 3
    fl = flops(doc, Nh, D)
    print(f"B{B} T{T} Nh{Nh}/{Nkv} D{D} docs={docs or 'causal'}: fwd {ms:.2f} ms {fl/ms/1e9:.0f} TFLOP/s ({fl/ms/1e9/2500:.2f}) | bwd {msb:.2f} ms {2.5*fl/msb/1e9:.0f} TFLOP/s ({2.5*fl/msb/1e9/2500:.2f})", flush=True)
