"""Per-kernel micro-benchmarks on the MI355X (run through gpurun): prints one JSON line per kernel with
the achieved rate against the roofline that bounds it (HBM 8 TB/s spec / bf16 MFMA 2.5 PF dense)."""
import json
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import touchnet_amd.functional as F  # noqa: E402

DEV = "cuda"
HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15
OUT = []


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rec(name, secs, bytes_=None, flops=None, **kw):
    r = {"kernel": name, "ms": round(secs * 1e3, 4)}
    if bytes_:
        r.update(GBps=round(bytes_ / secs / 1e9, 1), hbm_frac=round(bytes_ / secs / HBM_PEAK, 3))
    if flops:
        r.update(TFLOPs=round(flops / secs / 1e12, 1), mfma_frac=round(flops / secs / MFMA_PEAK, 3))
    r.update(kw)
    OUT.append(r)
    print(json.dumps(r), flush=True)


def docs(B, T, mean_len, seed=0):
    rng = np.random.RandomState(seed)
    out = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        t, d = 0, 1
        while t < T:
            n = max(1, int(rng.normal(mean_len, mean_len * 0.1)))
            out[b, t:t + n] = d
            t += n
            d += 1
    return torch.from_numpy(out)


def main():
    torch.manual_seed(0)
    print(torch.cuda.get_device_name(0), torch.__version__, flush=True)
    N, H, I, V = 16384, 4096, 11008, 156032
    bf = torch.bfloat16
    # ---- baseline: device copy
    a = torch.empty(N, I, dtype=bf, device=DEV).normal_()
    t = timeit(lambda: a.clone())
    rec("torch.clone bf16 [16384,11008]", t, bytes_=2 * a.numel() * 2)
    # ---- library GEMMs (hipBLASLt through torch) at the Qwen2-Audio-7B shapes
    x = torch.randn(N, H, dtype=bf, device=DEV)
    for name, (n_out, k_in) in {"qkv/o 4096x4096": (H, H), "gate/up 4096->11008": (I, H), "down 11008->4096": (H, I),
                                "lm_head 4096->156032": (V, H)}.items():
        w = torch.randn(n_out, k_in, dtype=bf, device=DEV) * 0.02
        xi = torch.randn(N, k_in, dtype=bf, device=DEV)
        dy = torch.randn(N, n_out, dtype=bf, device=DEV)
        fl = 2.0 * N * n_out * k_in
        rec(f"gemm fwd {name}", timeit(lambda: torch.nn.functional.linear(xi, w), 10, 3), flops=fl)
        rec(f"gemm dgrad {name}", timeit(lambda: dy @ w, 10, 3), flops=fl)
        rec(f"gemm wgrad {name}", timeit(lambda: dy.t() @ xi, 10, 3), flops=fl)
        del w, xi, dy
    # ---- HBM-bound kernels
    w = torch.ones(H, dtype=bf, device=DEV)
    r = torch.randn(N, H, dtype=bf, device=DEV)
    rec("rmsnorm fwd [16384,4096]", timeit(lambda: F.rms_norm(x, w, 1e-5)), bytes_=2 * N * H * 2)
    rec("add+rmsnorm fwd", timeit(lambda: F.rms_norm(x, w, 1e-5, residual=r)), bytes_=4 * N * H * 2)
    xg = x.clone().requires_grad_()
    wg = w.clone().requires_grad_()
    y = F.rms_norm(xg, wg, 1e-5)
    dy = torch.randn_like(y)
    rec("rmsnorm bwd", timeit(lambda: torch.autograd.grad(y, [xg, wg], dy, retain_graph=True)), bytes_=3 * N * H * 2)
    g_, u_ = torch.randn(N, I, dtype=bf, device=DEV, requires_grad=True), torch.randn(N, I, dtype=bf, device=DEV, requires_grad=True)
    rec("swiglu fwd [16384,11008]", timeit(lambda: F.swiglu(g_, u_)), bytes_=3 * N * I * 2)
    o = F.swiglu(g_, u_)
    do = torch.randn_like(o)
    rec("swiglu bwd", timeit(lambda: torch.autograd.grad(o, [g_, u_], do, retain_graph=True)), bytes_=5 * N * I * 2)
    del g_, u_, o, do
    B, T, Nh, D = 2, 8192, 32, 128
    q = torch.randn(B, T, Nh, D, dtype=bf, device=DEV)
    k = torch.randn(B, T, Nh, D, dtype=bf, device=DEV)
    pos = torch.arange(T, device=DEV).repeat(B, 1)
    inv = F.rope_inv_freq(D, 10000.0).to(DEV)
    cos, sin = F.rope_tables(pos, inv, bf)
    rec("rope q+k [2,8192,32,128]", timeit(lambda: F.apply_rope(q, k, cos, sin)), bytes_=4 * q.numel() * 2)
    # ---- cross entropy over the 156k vocabulary (all rows valid = worst case)
    for frac_valid in (1.0, 0.05):
        n = 4096
        logits = torch.randn(1, n, V, dtype=bf, device=DEV, requires_grad=True)
        labels = torch.randint(0, V, (1, n), device=DEV)
        labels[0, int(n * frac_valid):] = -100
        sl = torch.full((1, n), 7, device=DEV)
        nv = int(n * frac_valid)
        rec(f"CE fwd [4096,156032] valid={frac_valid}", timeit(lambda: F.packed_cross_entropy(logits, labels, sl, 10)),
            bytes_=nv * V * 2)
        loss, _ = F.packed_cross_entropy(logits, labels, sl, 10)
        rec(f"CE bwd valid={frac_valid}", timeit(lambda: torch.autograd.grad(loss, logits, retain_graph=True)),
            bytes_=(nv * 2 + (n - nv)) * V * 2)
        del logits, loss
    # ---- packed attention (Qwen2-Audio-7B heads) on ~790-token documents and on plain causal
    v = torch.randn(B, T, Nh, D, dtype=bf, device=DEV)
    for name, doc in {"docs~790": docs(B, T, 790), "causal": torch.ones(B, T, dtype=torch.int64),
                      "docs~100": docs(B, T, 100)}.items():
        mask = F.build_packed_mask(doc.to(DEV))
        allowed = 0
        for b in range(B):
            _, counts = np.unique(doc[b].numpy(), return_counts=True)
            allowed += int(sum(c * (c + 1) // 2 for c in counts))
        fl_f = 4.0 * D * Nh * allowed
        rec(f"attn fwd {name}", timeit(lambda: F.packed_attention(q, k, v, mask), 10, 3), flops=fl_f,
            formula_TFLOPs=round(4.0 * D * Nh * B * T * T / 1e12, 2))
        qg, kg, vg = [t.clone().requires_grad_() for t in (q, k, v)]
        o = F.packed_attention(qg, kg, vg, mask)
        do = torch.randn_like(o)
        rec(f"attn bwd {name}", timeit(lambda: torch.autograd.grad(o, [qg, kg, vg], do, retain_graph=True), 10, 3),
            flops=2.5 * fl_f)
        del qg, kg, vg, o, do
    # ---- Llama-1B heads (D=64, GQA 32:8)
    q = torch.randn(1, T, 32, 64, dtype=bf, device=DEV)
    k = torch.randn(1, T, 8, 64, dtype=bf, device=DEV)
    v = torch.randn(1, T, 8, 64, dtype=bf, device=DEV)
    doc = docs(1, T, 400)
    mask = F.build_packed_mask(doc.to(DEV))
    _, counts = np.unique(doc[0].numpy(), return_counts=True)
    fl_f = 4.0 * 64 * 32 * int(sum(c * (c + 1) // 2 for c in counts))
    rec("attn fwd D64 gqa docs~400", timeit(lambda: F.packed_attention(q, k, v, mask), 10, 3), flops=fl_f)
    qg, kg, vg = [t.clone().requires_grad_() for t in (q, k, v)]
    o = F.packed_attention(qg, kg, vg, mask)
    do = torch.randn_like(o)
    rec("attn bwd D64 gqa docs~400", timeit(lambda: torch.autograd.grad(o, [qg, kg, vg], do, retain_graph=True), 10, 3),
        flops=2.5 * fl_f)
    # ---- frontend
    wav = (torch.randn(16000 * 30, device=DEV) * 0.1).clamp(-1, 1)
    rec("kaldi fbank 30 s", timeit(lambda: F.kaldi_fbank(wav, 80)), bytes_=wav.numel() * 4 + 2998 * 80 * 4)
    rec("log-mel 30 s", timeit(lambda: F.log_mel_spectrogram(wav, 128)), bytes_=wav.numel() * 4 + 3000 * 128 * 4)
    json.dump(OUT, open("gpurun_out/microbench.json", "w"), indent=1)


if __name__ == "__main__":
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    main()
