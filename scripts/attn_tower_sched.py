import os, sys, torch
sys.path.insert(0, os.getcwd())
import touchnet_amd.functional as F
dev, bf = "cuda", torch.bfloat16
for (B, T, Nh, D) in ((20, 1500, 20, 64), (1, 65536, 32, 128)):
    q = torch.randn(B, T, Nh, D, dtype=bf, device=dev); k = torch.randn_like(q); v = torch.randn_like(q)
    mask = F.causal_mask(B, T, dev)
    with torch.no_grad():
        for _ in range(3): F.packed_attention(q, k, v, mask)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): F.packed_attention(q, k, v, mask)
        e.record(); torch.cuda.synchronize()
    print(os.environ.get("TN_ATTN_FWD_SCHEDULE", "0"), (B, T, Nh, D), round(s.elapsed_time(e) / 10, 3), "ms")
