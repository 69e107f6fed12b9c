"""GPU parity: every HIP kernel (through the C ABI / autograd wrappers) against the CPU oracle.

Tolerances: fp32 kernels 1e-5 rel; bf16 kernels are compared with the oracle evaluated in fp32 on the
same bf16-rounded inputs, tolerance = a few bf16 ulps of the output scale (stated per test)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import frontend as ofe
from oracle import loss as oloss
from oracle import nn as onn

DEV = "cuda"


def _f():
    import touchnet_amd.functional as F
    return F


def _ref(t):
    """fp32 CPU leaf (a real copy: `.float()` of an fp32 tensor would alias its input)."""
    return t.detach().float().clone().requires_grad_()


def _dev(t):
    return t.detach().clone().to(DEV).requires_grad_()


def _close(got, ref, atol, rtol, what):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} "
                           f"at {np.unravel_index(int(err.argmax()), err.shape)}, ref scale {float(ref.abs().max()):.4g}")


def test_library_loads_on_gpu():
    from touchnet_amd import _C
    assert "gfx950" in _C.version()
    assert torch.cuda.is_available()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,H", [(37, 64), (130, 256), (64, 1280), (96, 2048), (40, 3584), (33, 4096), (16, 8192)])
def test_rmsnorm(dtype, rows, H):   # 3584 = Kimi-Audio-7B hidden size (config E)
    F = _f()
    if dtype == torch.float32 and H > 4096:
        pytest.skip("fp32 rows wider than 4096 are outside the kernel's register budget (bf16 goes to 8192)")
    g = torch.Generator().manual_seed(rows * H)
    x = torch.randn(rows, H, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    dy = torch.randn(rows, H, generator=g).to(dtype)
    xr, wr = _ref(x), _ref(w)
    yr = onn.rms_norm(xr, wr, 1e-5)
    yr.backward(dy.float())
    xd, wd = _dev(x), _dev(w)
    y = F.rms_norm(xd, wd, 1e-5)
    y.backward(dy.to(DEV))
    a, rt = (1e-5, 1e-5) if dtype == torch.float32 else (3e-2, 2e-2)
    _close(y, yr, a, rt, "y")
    _close(xd.grad, xr.grad, a, rt, "dx")
    _close(wd.grad, wr.grad, a * math.sqrt(rows), rt, "dw")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rmsnorm_residual_grads(dtype):
    F = _f()
    rows, H = 70, 512
    g = torch.Generator().manual_seed(5)
    x, r = torch.randn(rows, H, generator=g).to(dtype), torch.randn(rows, H, generator=g).to(dtype)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    dy, dh = torch.randn(rows, H, generator=g).to(dtype), torch.randn(rows, H, generator=g).to(dtype)
    xr, rr, wr = _ref(x), _ref(r), _ref(w)
    h = xr + rr
    yr = onn.rms_norm(h, wr, 1e-5)
    torch.autograd.backward([yr, h], [dy.float(), dh.float()])
    xd, rd, wd = _dev(x), _dev(r), _dev(w)
    y, hh = F.rms_norm(xd, wd, 1e-5, residual=rd)
    torch.autograd.backward([y, hh], [dy.to(DEV), dh.to(DEV)])
    a, rt = (1e-5, 1e-5) if dtype == torch.float32 else (4e-2, 2e-2)
    _close(hh, h, a, rt, "h")
    _close(xd.grad, xr.grad, a, rt, "dx")
    _close(rd.grad, rr.grad, a, rt, "dres")
    _close(wd.grad, wr.grad, a * 8, rt, "dw")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,H", [(50, 32), (130, 1280)])
def test_layernorm(dtype, rows, H):
    F = _f()
    g = torch.Generator().manual_seed(3)
    x, r = torch.randn(rows, H, generator=g).to(dtype), torch.randn(rows, H, generator=g).to(dtype)
    w, b = (1 + 0.1 * torch.randn(H, generator=g)).to(dtype), (0.1 * torch.randn(H, generator=g)).to(dtype)
    dy, dh = torch.randn(rows, H, generator=g).to(dtype), torch.randn(rows, H, generator=g).to(dtype)
    xr, rr, wr, br = [_ref(t) for t in (x, r, w, b)]
    h = xr + rr
    yr = onn.layer_norm(h, wr, br)
    torch.autograd.backward([yr, h], [dy.float(), dh.float()])
    xd, rd, wd, bd = [_dev(t) for t in (x, r, w, b)]
    y, hh = F.layer_norm(xd, wd, bd, 1e-5, residual=rd)
    torch.autograd.backward([y, hh], [dy.to(DEV), dh.to(DEV)])
    a, rt = (2e-5, 1e-5) if dtype == torch.float32 else (4e-2, 2e-2)
    _close(y, yr, a, rt, "y")
    _close(xd.grad, xr.grad, a, rt, "dx")
    _close(wd.grad, wr.grad, a * 12, rt, "dw")
    _close(bd.grad, br.grad, a * 12, rt, "db")
    # no-residual variant
    y2 = F.layer_norm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5)
    _close(y2, onn.layer_norm(x.float(), w.float(), b.float()), a, rt, "y(no res)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_swiglu_gelu(dtype):
    F = _f()
    g = torch.Generator().manual_seed(2)
    a_, b_ = (torch.randn(77, 1024, generator=g) * 2).to(dtype), torch.randn(77, 1024, generator=g).to(dtype)
    d = torch.randn(77, 1024, generator=g).to(dtype)
    ar, br = _ref(a_), _ref(b_)
    onn.swiglu(ar, br).backward(d.float())
    ad, bd = _dev(a_), _dev(b_)
    out = F.swiglu(ad, bd)
    out.backward(d.to(DEV))
    at, rt = (1e-5, 1e-5) if dtype == torch.float32 else (2e-2, 2e-2)
    _close(out, onn.swiglu(a_.float(), b_.float()), at, rt, "swiglu")
    _close(ad.grad, ar.grad, at, rt, "dgate")
    _close(bd.grad, br.grad, at, rt, "dup")
    xr = _ref(a_)
    torch.nn.functional.gelu(xr).backward(d.float())
    xd = _dev(a_)
    o2 = F.gelu(xd)
    o2.backward(d.to(DEV))
    _close(o2, torch.nn.functional.gelu(a_.float()), at, rt, "gelu")
    _close(xd.grad, xr.grad, at, rt, "dgelu")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,hq,hk", [(8, 8, 4), (64, 4, 2), (128, 4, 4)])
def test_rope(golden, dtype, D, hq, hk):
    F = _f()
    scaling = dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                   original_max_position_embeddings=64)
    inv = F.rope_inv_freq(D, 500000.0, scaling)
    np.testing.assert_allclose(inv.numpy(), onn.rope_inv_freq(D, 500000.0, scaling).numpy(), rtol=1e-6)
    B, T = 2, 50
    g = torch.Generator().manual_seed(1)
    pos = torch.stack([torch.cat([torch.arange(20), torch.arange(25), torch.zeros(5, dtype=torch.int64)]),
                       torch.arange(50)])
    q, k = torch.randn(B, T, hq, D, generator=g).to(dtype), torch.randn(B, T, hk, D, generator=g).to(dtype)
    dq, dk = torch.randn(B, T, hq, D, generator=g).to(dtype), torch.randn(B, T, hk, D, generator=g).to(dtype)
    cos, sin = F.rope_tables(pos.to(DEV), inv.to(DEV), dtype)
    cr, sr = onn.rope_cos_sin(pos, inv, torch.float32)
    at = 1e-5 if dtype == torch.float32 else 1e-2
    _close(cos, cr[..., :D // 2].reshape(B * T, -1), at, 0, "cos")
    _close(sin, sr[..., :D // 2].reshape(B * T, -1), at, 0, "sin")
    qr, kr = _ref(q), _ref(k)
    cr2, sr2 = onn.rope_cos_sin(pos, inv, dtype)       # the eager path casts the table to the activation dtype
    qo, ko = onn.apply_rope(qr.transpose(1, 2), kr.transpose(1, 2), cr2.float(), sr2.float())
    torch.autograd.backward([qo, ko], [dq.float().transpose(1, 2), dk.float().transpose(1, 2)])
    qd, kd = _dev(q), _dev(k)
    qo2, ko2 = F.apply_rope(qd, kd, cos, sin)
    torch.autograd.backward([qo2, ko2], [dq.to(DEV), dk.to(DEV)])
    at, rt = (1e-5, 1e-5) if dtype == torch.float32 else (3e-2, 1e-2)
    _close(qo2, qo.transpose(1, 2), at, rt, "q")
    _close(ko2, ko.transpose(1, 2), at, rt, "k")
    _close(qd.grad, qr.grad, at, rt, "dq")
    _close(kd.grad, kr.grad, at, rt, "dk")


@pytest.mark.parametrize("case", ["pack", "mixed"])
def test_ce_golden(golden, case):
    F = _f()
    g = golden("ce_loss.npz")
    logits = torch.tensor(g[f"{case}/logits"]).to(DEV).requires_grad_()
    labels, sl = torch.tensor(g[f"{case}/labels"]).to(DEV), torch.tensor(g[f"{case}/sentence_lens"]).to(DEV)
    loss, stats = F.packed_cross_entropy(logits, labels, sl, int(g[f"{case}/num_sentence"]))
    loss.backward()
    assert float(stats[0]) == pytest.approx(float(g[f"{case}/loss_per_sample"]), abs=1e-6)
    assert float(stats[1]) == pytest.approx(float(g[f"{case}/loss_per_token"]), abs=1e-6)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), g[f"{case}/dlogits"], atol=1e-7)
    acc = oloss.accuracy(torch.tensor(g[f"{case}/logits"]), torch.tensor(g[f"{case}/labels"]))
    assert float(stats[2]) == pytest.approx(float(acc), abs=1e-7)


@pytest.mark.parametrize("dtype,V", [(torch.float32, 1000), (torch.bfloat16, 128256), (torch.bfloat16, 156032),
                                      (torch.bfloat16, 168448), (torch.bfloat16, 1003)])   # 168448 = Kimi-Audio (config E)
def test_ce_large_vocab(dtype, V):
    F = _f()
    B, T = 2, 24
    g = torch.Generator().manual_seed(V)
    logits = (torch.randn(B, T, V, generator=g) * 2).to(dtype)
    logits[0, 3, 17] = logits[0, 3].max() + 1          # a clear argmax
    logits[1, 2, 5] = logits[1, 2, 900] = logits[1, 2].float().max() + 2   # a tie: first index wins
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[0, 3], labels[1, 2] = 17, 5
    labels[0, :5] = -100
    labels[0, 3] = 17
    labels[1, 20:] = -100
    sl = torch.randint(1, 9, (B, T), generator=g)
    lr = _ref(logits)
    ps, pt = oloss.cross_entropy_loss(lr, labels, sl, 7)
    (ps * 0.5).backward()
    ld = _dev(logits)
    loss, stats = F.packed_cross_entropy(ld, labels.to(DEV), sl.to(DEV), torch.tensor(7, device=DEV))
    (loss * 0.5).backward()
    assert float(stats[0]) == pytest.approx(float(ps), rel=2e-6)
    assert float(stats[1]) == pytest.approx(float(pt), rel=2e-6)
    assert float(stats[2]) == pytest.approx(float(oloss.accuracy(logits.float(), labels)), abs=1e-7)
    assert float(stats[3]) == float((labels != -100).sum())
    at = 1e-8 if dtype == torch.float32 else 2e-4
    _close(ld.grad, lr.grad, at, 1e-2 if dtype == torch.bfloat16 else 1e-5, "dlogits")
    assert float(ld.grad[0, 0].abs().max()) == 0.0     # ignored rows are zero-filled


def test_ce_all_ignored():
    F = _f()
    logits = torch.randn(1, 4, 16, device=DEV, requires_grad=True)
    loss, stats = F.packed_cross_entropy(logits, torch.full((1, 4), -100, device=DEV),
                                         torch.ones(1, 4, dtype=torch.int64, device=DEV), 1)
    loss.backward()
    assert stats.tolist() == [0.0, 0.0, 0.0, 0.0] and float(logits.grad.abs().max()) == 0.0


def _docs(B, T, seed, maxlen, pad_tail):
    rng = np.random.RandomState(seed)
    out = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        t, d = 0, 1
        end = T - (int(rng.randint(0, pad_tail + 1)) if pad_tail else 0)
        while t < end:
            n = int(rng.randint(1, maxlen + 1))
            out[b, t:min(t + n, end)] = d
            t += n
            d += 1
    return torch.from_numpy(out)


ATT_CASES = [
    # B, T, Nh, Nkv, D, maxdoc, pad
    (2, 128, 2, 2, 64, 50, 10),
    (1, 256, 4, 2, 128, 100, 30),
    (2, 200, 4, 1, 64, 70, 20),        # T not a multiple of 64, GQA 4:1
    (1, 512, 2, 2, 128, 512, 0),       # one or two long docs
    (1, 1500, 2, 2, 64, 1500, 0),      # whisper-tower length, plain causal (single doc)
    (2, 384, 8, 4, 128, 7, 5),         # many tiny docs
    (1, 1024, 4, 4, 128, 300, 100),
    (1, 640, 14, 2, 128, 200, 17),     # Kimi-Audio head geometry (28 q heads / 4 kv heads: GQA 7:1), halved
]


@pytest.mark.parametrize("B,T,Nh,Nkv,D,maxdoc,pad", ATT_CASES)
def test_packed_attention_fwd_bwd(B, T, Nh, Nkv, D, maxdoc, pad):
    F = _f()
    doc = _docs(B, T, T + Nh, maxdoc, pad)
    g = torch.Generator().manual_seed(T)
    q = torch.randn(B, T, Nh, D, generator=g).bfloat16()
    k = torch.randn(B, T, Nkv, D, generator=g).bfloat16()
    v = torch.randn(B, T, Nkv, D, generator=g).bfloat16()
    do = torch.randn(B, T, Nh, D, generator=g).bfloat16()
    qr, kr, vr = [_ref(t) for t in (q, k, v)]
    ref = onn.attention(qr.transpose(1, 2), kr.transpose(1, 2), vr.transpose(1, 2), onn.doc_causal_allow(doc),
                        D ** -0.5)
    ref.backward(do.float())
    qd, kd, vd = [_dev(t) for t in (q, k, v)]
    mask = F.build_packed_mask(doc.to(DEV))
    out = F.packed_attention(qd, kd, vd, mask)
    out.backward(do.to(DEV))
    torch.cuda.synchronize()
    # bf16 output of O(1) values: 2 bf16 ulps ~ 1.6e-2; P is rounded to bf16 before P.V (as flash kernels do)
    _close(out, ref, 2e-2, 2e-2, "O")
    pad_rows = (doc == 0)
    assert float(out.float().cpu()[pad_rows].abs().max() if pad_rows.any() else 0.0) == 0.0
    _close(vd.grad, vr.grad, 4e-2, 3e-2, "dV")
    _close(kd.grad, kr.grad, 4e-2, 3e-2, "dK")
    _close(qd.grad, qr.grad, 4e-2, 3e-2, "dQ")
    if pad_rows.any():
        assert float(qd.grad.float().cpu()[pad_rows].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,Nh,Nkv,D,maxdoc,pad", ATT_CASES + [(1, 2048, 4, 4, 128, 800, 60)])
def test_attention_backward_with_the_rotary_gradient_in_its_epilogues(B, T, Nh, Nkv, D, maxdoc, pad):
    """tn_attn_bwd_rope (functional.packed_attention(rope_grad=..) -> library.attn_bwd_rope): dq / dk come back as gradients
    of the UN-rotated projections — the bits of tn_attn_bwd followed by tn_rope_apply(backward) (D = 128: the transposed
    rotation sits in the dQ and dK / dV kernels' epilogues; D = 64: the row kernel runs behind the launch), dv unchanged.
    Multi-head and grouped-query geometries (the stacked and the flat result buffer), ragged T, padded tails."""
    F = _f()
    import touchnet_amd.library as L
    doc = _docs(B, T, T + Nh, maxdoc, pad)
    g = torch.Generator().manual_seed(T + D)
    q, do = [torch.randn(B, T, Nh, D, generator=g).bfloat16().to(DEV) for _ in range(2)]
    k, v = [torch.randn(B, T, Nkv, D, generator=g).bfloat16().to(DEV) for _ in range(2)]
    pos = torch.randint(0, 4000, (B, T), generator=g)
    cos, sin = F.rope_tables(pos.to(DEV), F.rope_inv_freq(D, 1000000.0, device=DEV), torch.bfloat16)
    mask = F.build_packed_mask(doc.to(DEV))
    qa, ka, va = [t.clone().requires_grad_(True) for t in (q, k, v)]
    F.packed_attention(qa, ka, va, mask).backward(do)
    rq, rk = L.rope_apply(qa.grad, ka.grad, cos, sin, True)
    qb, kb, vb = [t.clone().requires_grad_(True) for t in (q, k, v)]
    F.packed_attention(qb, kb, vb, mask, rope_grad=(cos, sin)).backward(do)
    torch.cuda.synchronize()
    assert torch.equal(vb.grad, va.grad)
    assert torch.equal(qb.grad, rq), float((qb.grad.float() - rq.float()).abs().max())
    assert torch.equal(kb.grad, rk), float((kb.grad.float() - rk.float()).abs().max())


@pytest.mark.parametrize("D", [128, 64])
def test_packed_attention_backward_is_reproducible_launch_after_launch(D):
    """The same backward 300 times on a batch of short documents (the shape of the 2560-wide trainer test: B = 4 x 512,
    documents <= 90 tokens, 20 heads): every launch returns the bits of the first one.  Round 4's fused dK + dV kernel
    read the four waves' row statistics from LDS without a barrier behind their stores; about one launch in 300 dropped
    or mis-masked stages (seen as an irreproducible grad norm in a trainer test, never by the tolerance tests above)."""
    F = _f()
    B, T, Nh = 4, 512, 20
    doc = _docs(B, T, 5, 90, 0)
    g = torch.Generator().manual_seed(D)
    q, k, v, do = [torch.randn(B, T, Nh, D, generator=g).bfloat16().to(DEV) for _ in range(4)]
    mask = F.build_packed_mask(doc.to(DEV))
    ref = None
    for it in range(300):
        qd, kd, vd = [t.clone().requires_grad_(True) for t in (q, k, v)]
        F.packed_attention(qd, kd, vd, mask).backward(do)
        cur = [qd.grad, kd.grad, vd.grad]
        if ref is None:
            ref = cur
            continue
        for name, a, b in zip(("dQ", "dK", "dV"), cur, ref):
            assert torch.equal(a, b), (it, name, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("B,T,Nh,Nkv,D,maxdoc,pad", [
    (2, 300, 4, 4, 64, 300, 0), (1, 512, 4, 2, 128, 200, 40), (3, 1500, 2, 2, 64, 1500, 0),     # 1500: one Whisper clip
    (2, 384, 8, 4, 128, 7, 5),          # many tiny documents
    (2, 200, 4, 1, 64, 70, 20),         # T not a multiple of 64, GQA 4:1
    (1, 1024, 4, 4, 64, 300, 100), (1, 640, 2, 2, 128, 640, 0)])
def test_bidirectional_attention_kernels(B, T, Nh, Nkv, D, maxdoc, pad):
    """functional.bidirectional_attention (Whisper's encoder self-attention for Kimi-Audio's speech encoder):
    tn_attn_fwd_bidir / tn_attn_bwd_bidir — the packed kernels without the causal term, key range up to the last tile of
    the document.  Against the fp32 oracle (softmax over all same-document keys): output, dQ, dK, dV; pad rows exactly 0."""
    import oracle.ops as oops
    F = _f()
    doc = _docs(B, T, T + Nh, maxdoc, pad)
    g = torch.Generator().manual_seed(T + 1)
    q = torch.randn(B, T, Nh, D, generator=g).bfloat16()
    k = torch.randn(B, T, Nkv, D, generator=g).bfloat16()
    v = torch.randn(B, T, Nkv, D, generator=g).bfloat16()
    do = torch.randn(B, T, Nh, D, generator=g).bfloat16()
    qr, kr, vr = [_ref(t) for t in (q, k, v)]
    ref = oops.bidirectional_attention(qr, kr, vr, oops.build_packed_mask(doc))
    ref.backward(do.float())
    qd, kd, vd = [_dev(t) for t in (q, k, v)]
    out = F.bidirectional_attention(qd, kd, vd, F.build_packed_mask(doc.to(DEV)))
    out.backward(do.to(DEV))
    torch.cuda.synchronize()
    _close(out, ref, 2e-2, 2e-2, "O")
    pad_rows = (doc == 0)
    if pad_rows.any():
        assert float(out.float().cpu()[pad_rows].abs().max()) == 0.0
        assert float(qd.grad.float().cpu()[pad_rows].abs().max()) == 0.0
    _close(vd.grad, vr.grad, 4e-2, 3e-2, "dV")
    _close(kd.grad, kr.grad, 4e-2, 3e-2, "dK")
    _close(qd.grad, qr.grad, 4e-2, 3e-2, "dQ")


def test_packed_attention_online_softmax_rescale():
    """Force a large running-max jump at a late KV tile (guide rule 26: the rescale branch needs its own test)."""
    F = _f()
    B, T, Nh, D = 1, 384, 1, 128
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B, T, Nh, D, generator=g) * 0.3).bfloat16()
    k = (torch.randn(B, T, Nh, D, generator=g) * 0.3).bfloat16()
    v = torch.randn(B, T, Nh, D, generator=g).bfloat16()
    k[0, 300, 0] = q[0, 350, 0] * 40        # spike: row 350 sees its max jump at KV tile 4 (forces the rescale)
    k[0, 200, 0] = q[0, 260, 0] * 3         # mild growth (< 2^8): exercises the deferred-rescale path
    doc = torch.ones(B, T, dtype=torch.int64)
    ref = onn.attention(q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2),
                        onn.doc_causal_allow(doc), D ** -0.5)
    out = F.packed_attention(q.to(DEV), k.to(DEV), v.to(DEV), F.build_packed_mask(doc.to(DEV)))
    _close(out, ref, 2e-2, 2e-2, "O with spike")


def test_packed_mask_metadata_matches_predicate(golden):
    """Block-sparsity metadata must never drop an allowed (q, kv) pair (bit-exact predicate coverage)."""
    F = _f()
    g = golden("docmask.npz")
    doc = torch.tensor(g["big/doc_ids"])
    B, T = doc.shape
    m = F.build_packed_mask(doc.to(DEV))
    nt = (T + 63) // 64
    meta = m.meta.cpu().numpy()[:5 * B * nt].reshape(5, B, nt)
    allow = g["big/allow"]
    for b in range(B):
        for qt in range(nt):
            lo, = meta[3, b, qt:qt + 1]
            rows = allow[b, qt * 64:(qt + 1) * 64]
            used = np.nonzero(rows.any(0))[0]
            if used.size:
                assert lo <= used.min() // 64, (b, qt, lo, used.min())
        for kt in range(nt):
            hi, = meta[4, b, kt:kt + 1]
            cols = allow[b, :, kt * 64:(kt + 1) * 64]
            used = np.nonzero(cols.any(1))[0]
            if used.size:
                assert hi >= used.max() // 64, (b, kt, hi, used.max())


def _meta_cases(golden):
    """(doc ids [B, T], allow [B, T, T]) pairs: the reference-run mask of docmask.npz, packed batches with padded tails and a
    ragged T, one long document (lists longer than the stored 64 entries), ids that are not runs in increasing order"""
    g = golden("docmask.npz")
    yield torch.tensor(g["big/doc_ids"]), g["big/allow"]
    for B, T, maxlen, pad, seed in ((2, 1000, 150, 40, 1), (1, 2048, 700, 0, 2), (3, 333, 40, 30, 3), (1, 9000, 9000, 0, 4)):
        doc = _docs(B, T, seed, maxlen, pad)
        yield doc, onn.doc_causal_allow(doc).numpy()
    doc = torch.tensor([[3] * 100 + [1] * 200 + [0] * 10 + [2] * 150 + [1] * 60 + [0] * 20])
    yield doc, onn.doc_causal_allow(doc).numpy()


def test_round6_tile_lists_and_row_statistics_never_drop_an_allowed_pair(golden):
    """The round-6 metadata of tn_attn_build_meta (csrc/attn_common.h: qstat per 32 rows, klist per 128-row query tile, qlist
    per 128-row KV tile — what the stream kernels and the dK / dV passes read instead of deriving it): the statistics are
    EXACT, the lists are sorted, duplicate-free, carry the tiles' own statistics, and contain every tile that holds an
    allowed (q, kv) pair of the reference's predicate (a list longer than the 64 stored entries says so by its count: the
    kernels then build it themselves)."""
    F = _f()
    K = 64                                                              # kListPre
    for doc, allow in _meta_cases(golden):
        B, T = doc.shape
        nt, nq32, nq128 = (T + 63) // 64, (T + 31) // 32, (T + 127) // 128
        meta = F.build_packed_mask(doc.to(DEV)).meta.cpu().numpy()
        tile = meta[:5 * B * nt].reshape(5, B, nt)                      # tmin, tmax, tminpos, q_lo, kv_hi
        o = 5 * B * nt
        qstat = meta[o:o + 4 * B * nq32].reshape(B, nq32, 4)
        o += 4 * B * nq32
        rec = 4 + 4 * K
        klist = meta[o:o + rec * B * nq128].reshape(B, nq128, 1 + K, 4)
        qlist = meta[o + rec * B * nq128:o + 2 * rec * B * nq128].reshape(B, nq128, 1 + K, 4)
        d = doc.numpy()
        for b in range(B):
            for w in range(nq32):
                rows = d[b, 32 * w:32 * w + 32]
                pos = rows[rows > 0]
                want = (int(pos.min()) if pos.size else 0x7fffffff, int(rows.max()) if rows.size else 0,
                        int((rows == 0).any() or rows.size < 32))
                assert tuple(int(v) for v in qstat[b, w, :3]) == want, (b, w, qstat[b, w], want)
            for lists, q_side in ((klist, True), (qlist, False)):
                for t128 in range(nq128):
                    count, own = int(lists[b, t128, 0, 0]), int(lists[b, t128, 0, 1])
                    assert own == t128
                    blk = allow[b, 128 * t128:128 * t128 + 128] if q_side else allow[b, :, 128 * t128:128 * t128 + 128]
                    used = blk.any(0) if q_side else blk.any(1)                       # positions on the other side
                    need = sorted({int(p) // 64 for p in np.nonzero(used)[0]})
                    if count > K:                                                     # overflow: the kernels rebuild the list
                        assert len(need) <= count
                        continue
                    ent = lists[b, t128, 1:1 + count]
                    got = [int(v) for v in ent[:, 0]]
                    assert got == sorted(set(got)), (b, t128, got)
                    assert set(need) <= set(got), (b, t128, q_side, need, got)
                    for j, mn, mx, mp in ent:                                         # each entry carries its tile's statistics
                        assert (mn, mx, mp) == tuple(tile[:3, b, j]), (b, t128, j)


def test_frontend_stack(golden):
    F = _f()
    g = golden("audiofeat_stack.npz")
    for n in sorted({k.split("/")[1] for k in g.files}):
        stack, stride = [int(v) for v in g[f"stack/{n}/ss"]]
        y = F.audiofeat_stack(torch.tensor(g[f"stack/{n}/x"]).to(DEV), stack, stride).cpu().numpy()
        assert y.shape == g[f"stack/{n}/y"].shape, n
        if g[f"stack/{n}/x"].shape[1] * stack > 1:
            np.testing.assert_allclose(y, g[f"stack/{n}/y"], atol=5e-5, err_msg=n)


def test_feature_augmentation_on_the_device_replays_the_reference_runs(golden):
    """spec_aug -> spec_sub -> spec_trim of the reference's datapipe (functions.py:193-255) through the product stage
    functions with the HIP gather kernel: bit-identical to the outputs of the reference's own stage functions under the
    same seed — one fused launch per utterance, and the three standalone stages chained."""
    import random
    import types
    from touchnet_amd.data import functions as fn
    keys = ("audiofeat_spec_aug", "audiofeat_spec_aug_num_t_mask", "audiofeat_spec_aug_num_f_mask",
            "audiofeat_spec_aug_max_t", "audiofeat_spec_aug_max_f", "audiofeat_spec_sub", "audiofeat_spec_sub_num_t_sub",
            "audiofeat_spec_sub_max_t", "audiofeat_spec_trim", "audiofeat_spec_trim_max_t")
    g = golden("audiofeat_augment.npz")
    for n in [str(v) for v in g["names"]]:
        cfg = types.SimpleNamespace(**{k: int(v) for k, v in zip(keys, g[f"{n}/cfg"])})
        k = int(g[f"{n}/n"])
        for fused in (True, False):
            data = iter([{"audiofeat": torch.tensor(g[f"{n}/x{i}"]).to(DEV)} for i in range(k)])
            if fused:
                data = fn.audiofeat_augment(data, cfg)
            else:
                for flag, stage in (("audiofeat_spec_aug", fn.audiofeat_spec_aug), ("audiofeat_spec_sub", fn.audiofeat_spec_sub),
                                    ("audiofeat_spec_trim", fn.audiofeat_spec_trim)):
                    if getattr(cfg, flag):
                        data = stage(data, cfg)
            random.seed(int(g[f"{n}/seed"]))
            for i, smp in enumerate(data):
                y = smp["audiofeat"]
                assert y.is_cuda and tuple(y.shape) == g[f"{n}/y{i}"].shape, (n, i, fused)
                assert np.array_equal(y.cpu().numpy(), g[f"{n}/y{i}"]), (n, i, fused)
    # the C ABI refuses what it cannot represent
    F = _f()
    x = torch.randn(8, 4, device=DEV)
    with pytest.raises(Exception):
        F.feat_augment(x, subs=[(2, 4, 3)])                 # pos > start: the source row would lie in front of x
    with pytest.raises(Exception):
        F.feat_augment(x, t_masks=[(0, 1)] * 17)


def test_speed_perturbation_on_the_device():
    """functional.speed_perturb (polyphase table, fp32) against the oracle's float64 evaluation of the same band-limited
    interpolation, directly from the formula: speech-like noise and a tone, both recipe speeds, an odd length; through the
    datapipe stage too (int16 PCM in, the draw of `random.choice` decides)."""
    import random
    import types
    from oracle import frontend as ofe
    from touchnet_amd.data import functions as fn
    F = _f()
    rng = np.random.RandomState(3)
    for n in (16000, 4801):
        x = (rng.randn(n) * 0.1).astype(np.float32)
        x[: n // 2] += np.sin(2 * np.pi * 700 * np.arange(n // 2) / 16000).astype(np.float32) * 0.5
        for speed in (0.9, 1.1):
            y = F.speed_perturb(torch.from_numpy(x).to(DEV), speed).cpu().numpy()
            ref = ofe.speed_perturb(x, speed)
            assert y.shape == ref.shape
            assert np.abs(y - ref).max() < 5e-6, (n, speed, np.abs(y - ref).max())
    assert F.speed_perturb(torch.from_numpy(x).to(DEV), 1.0).data_ptr() != 0
    cfg = types.SimpleNamespace(audio_speed_perturb_speeds=[0.9, 1.0, 1.1])
    pcm = (rng.randn(1, 8000) * 3000).astype(np.int16)
    random.seed(11)
    picks = [random.choice(cfg.audio_speed_perturb_speeds) for _ in range(6)]
    random.seed(11)
    outs = list(fn.audio_speed_perturb(iter([{"sample_rate": 16000, "waveform": torch.from_numpy(pcm.copy())} for _ in range(6)]), cfg))
    for sp, smp in zip(picks, outs):
        w = smp["waveform"]
        assert w.shape[1] == (8000 if sp == 1.0 else 8000 * 10 // round(sp * 10))
        if sp != 1.0:
            assert w.is_cuda and w.dtype == torch.float32
            ref = ofe.speed_perturb(pcm[0].astype(np.float32) / 32768.0, sp)
            assert np.abs(w.cpu().numpy()[0] - ref).max() < 5e-6


def test_frontend_log_mel(golden):
    F = _f()
    g = golden("logmel.npz")
    np.testing.assert_allclose(F.slaney_mel_filters(128, "cpu").numpy(), g["mel_filters_128"], atol=1e-7)
    for i in (0, 1):
        wav = torch.tensor(g[f"wav{i}/pcm"].astype(np.float32) / 32768.0).to(DEV)
        for n_mels in (80, 128):
            y = F.log_mel_spectrogram(wav, n_mels).cpu().numpy()
            ref = g[f"wav{i}/logmel{n_mels}"]
            assert y.shape == ref.shape
            np.testing.assert_allclose(y, ref, atol=1e-3)


def test_frontend_kaldi_fbank(golden):
    """Against the oracle restatement on random clips and against the third-party Kaldi-compatible fbank fixture
    (transformers.audio_utils on the reference's two test wavs, tests/golden/kaldi_fbank_hf.npz); torchaudio itself is
    not available in this image — and against Kaldi's pipeline written from its specification (tests/golden/kaldi_from_spec.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import kaldi_from_spec as ks
    F = _f()
    g, wavs = golden("kaldi_fbank_hf.npz"), golden("logmel.npz")
    for i in (0, 1):
        pcm = torch.from_numpy(wavs[f"wav{i}/pcm"].astype(np.int16)).to(DEV)
        y = F.kaldi_fbank(F.pcm16_to_float(pcm), 80).cpu().numpy()
        assert y.shape == g[f"wav{i}/fbank80"].shape
        np.testing.assert_allclose(y, g[f"wav{i}/fbank80"], atol=3e-3)
        np.testing.assert_allclose(y, ks.reference_recipe_fbank(pcm.cpu().numpy(), 80), atol=3e-3)   # from-spec Kaldi (fp64)
    rng = np.random.RandomState(0)
    for n in (400, 16000, 16000 * 3 + 77):
        wav = np.clip(rng.randn(n) * 0.1, -1, 1).astype(np.float32)
        y = F.kaldi_fbank(torch.tensor(wav).to(DEV), 80).cpu().numpy()
        ref = ofe.kaldi_fbank(wav, 16000, 80)
        assert y.shape == ref.shape
        np.testing.assert_allclose(y, ref, atol=2e-3)
    assert F.kaldi_fbank(torch.zeros(399, device=DEV), 80).shape == (0, 80)


# ---------------------------------------------------------------------------------------------------------
# BASELINE-size checks through size-independent properties (the oracle cannot finish these sizes in seconds)
# ---------------------------------------------------------------------------------------------------------
def test_attention_full_size_packed_equals_per_document():
    """T = 8192, Qwen2-Audio head shape: attention over the packed row == attention over each document run on
    its own (the pack-vs-pad equivalence tests/touchnet/utils/test_pack_loss.py asserts for the loss), forward
    and backward, plus exact zeros on pad rows."""
    F = _f()
    B, T, Nh, Nkv, D = 1, 8192, 8, 8, 128
    doc = _docs(B, T, 77, 1400, 300)
    g = torch.Generator().manual_seed(8)
    q, k, v, do = [torch.randn(B, T, n, D, generator=g).bfloat16().to(DEV) for n in (Nh, Nkv, Nkv, Nh)]
    qg, kg, vg = [t.clone().requires_grad_() for t in (q, k, v)]
    out = F.packed_attention(qg, kg, vg, F.build_packed_mask(doc.to(DEV)))
    out.backward(do)
    ids = doc[0].numpy()
    for d in np.unique(ids[ids > 0])[:4]:
        idx = np.nonzero(ids == d)[0]
        s, e = int(idx[0]), int(idx[-1]) + 1
        q1, k1, v1 = [t[:, s:e].clone().requires_grad_() for t in (q, k, v)]
        o1 = F.packed_attention(q1, k1, v1, F.causal_mask(1, e - s, DEV))
        o1.backward(do[:, s:e])
        _close(out[:, s:e], o1, 1e-2, 1e-2, f"O doc {d}")
        _close(qg.grad[:, s:e], q1.grad, 2e-2, 2e-2, f"dQ doc {d}")
        _close(kg.grad[:, s:e], k1.grad, 2e-2, 2e-2, f"dK doc {d}")
        _close(vg.grad[:, s:e], v1.grad, 2e-2, 2e-2, f"dV doc {d}")
    pad = torch.from_numpy(ids == 0)
    assert float(out[0][pad].float().abs().max()) == 0.0 and float(qg.grad[0][pad].float().abs().max()) == 0.0
    assert float(kg.grad[0][pad].float().abs().max()) == 0.0


@pytest.mark.parametrize("T,Nh,D", [(1024, 4, 128), (2048, 8, 64), (4096, 16, 64)])
def test_attention_deterministic(T, Nh, D):
    """Run-to-run bit identity, 4 repetitions on enough workgroups to fill the chip (a hazard between an MFMA and an
    inline-asm consumer once showed up ONLY as 1-ulp run-to-run differences of the D = 64 forward)."""
    F = _f()
    B = 2
    doc = _docs(B, T, 5, 300, 40).to(DEV)
    g = torch.Generator().manual_seed(1)
    q, k, v, do = [torch.randn(B, T, Nh, D, generator=g).bfloat16().to(DEV) for _ in range(4)]
    res = []
    for _ in range(4):
        qg, kg, vg = [t.clone().requires_grad_() for t in (q, k, v)]
        o = F.packed_attention(qg, kg, vg, F.build_packed_mask(doc))
        o.backward(do)
        res.append((o.detach(), qg.grad, kg.grad, vg.grad))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)      # bit-identical: no atomics anywhere (recompute under AC is safe)


def test_ce_full_vocab_properties():
    """V = 156032 (Qwen2-Audio): loss is additive over a split of the packed row (the CP / sequence-shard
    property of test_pack_loss.py) and the fused lm_head+CE equals lm_head followed by CE."""
    F = _f()
    V, H, n = 156032, 256, 96
    g = torch.Generator().manual_seed(4)
    h = torch.randn(1, n, H, generator=g).bfloat16().to(DEV)
    w = (torch.randn(V, H, generator=g) * 0.05).bfloat16().to(DEV)
    labels = torch.randint(0, V, (1, n), generator=g).to(DEV)
    labels[0, ::3] = -100
    sl = torch.randint(1, 20, (1, n), generator=g).to(DEV)
    logits = torch.nn.functional.linear(h, w)
    full, st_full = F.packed_cross_entropy(logits, labels, sl, 9)
    a, _ = F.packed_cross_entropy(logits[:, :40], labels[:, :40], sl[:, :40], 9)
    b, _ = F.packed_cross_entropy(logits[:, 40:], labels[:, 40:], sl[:, 40:], 9)
    assert float(a + b) == pytest.approx(float(full), rel=1e-5)
    hg, wg = h.clone().requires_grad_(), w.clone().requires_grad_()
    l1, st1 = F.fused_linear_cross_entropy(hg, wg, labels, sl, 9, -100, 32)       # 3 chunks
    l1.backward()
    h2, w2 = h.clone().requires_grad_(), w.clone().requires_grad_()
    l2, st2 = F.packed_cross_entropy(torch.nn.functional.linear(h2, w2), labels, sl, 9)
    l2.backward()
    assert float(l1) == pytest.approx(float(l2), rel=1e-5) and float(st1[2]) == float(st2[2])
    _close(hg.grad, h2.grad, 2e-4, 2e-2, "dh")
    _close(wg.grad, w2.grad, 2e-4, 3e-2, "dW")
    # opt-in row compaction (lm_head only on labelled positions): same loss, stats and gradients; ignored rows get 0
    h3, w3 = h.clone().requires_grad_(), w.clone().requires_grad_()
    l3, st3 = F.fused_linear_cross_entropy(h3, w3, labels, sl, 9, -100, 32, compact=True)
    l3.backward()
    assert float(l3) == pytest.approx(float(l1), rel=1e-6) and torch.allclose(st3, st1, rtol=1e-6, atol=0)
    _close(h3.grad, hg.grad, 1e-3, 1e-2, "dh compact")           # (different chunking -> summation order: bf16 ulps)
    _close(w3.grad, wg.grad, 1e-3, 1e-2, "dW compact")
    assert float(h3.grad[0, ::3].abs().max()) == 0.0
    none = torch.full_like(labels, -100)
    l4, _ = F.fused_linear_cross_entropy(h.clone().requires_grad_(), w, none, sl, 9, -100, 32, compact=True)
    assert float(l4) == 0.0


@pytest.mark.parametrize("cp,T,Nh,Nkv,D,maxdoc", [(2, 1024, 4, 2, 128, 300), (4, 2048, 2, 2, 64, 2048),
                                                  (2, 512, 4, 4, 128, 40)])
def test_sharded_attention_emulated_context_parallel(cp, T, Nh, Nkv, D, maxdoc):
    """Context-parallel kernels on ONE GPU: every emulated rank runs its head/tail query shard against the full
    K/V; local outputs / dQ must equal the rows of the full-sequence kernel, partial dK/dV must SUM to the full
    dK/dV (what the reduce-scatter does)."""
    F = _f()
    B = 2
    doc = _docs(B, T, 3 + cp, maxdoc, 30).to(DEV)
    g = torch.Generator().manual_seed(T + cp)
    q = torch.randn(B, T, Nh, D, generator=g).bfloat16().to(DEV)
    k = torch.randn(B, T, Nkv, D, generator=g).bfloat16().to(DEV)
    v = torch.randn(B, T, Nkv, D, generator=g).bfloat16().to(DEV)
    do = torch.randn(B, T, Nh, D, generator=g).bfloat16().to(DEV)
    mask = F.build_packed_mask(doc)
    qf, kf, vf = [t.clone().requires_grad_() for t in (q, k, v)]
    of = F.packed_attention(qf, kf, vf, mask)
    of.backward(do)
    Tc = T // (2 * cp)
    dk_sum, dv_sum = torch.zeros_like(k, dtype=torch.float32), torch.zeros_like(v, dtype=torch.float32)
    for r in range(cp):
        pos = torch.cat([torch.arange(r * Tc, (r + 1) * Tc), torch.arange((2 * cp - 1 - r) * Tc, (2 * cp - r) * Tc)])
        shard = F.SeqShard(((0, Tc, r * Tc), (Tc, Tc, (2 * cp - 1 - r) * Tc)), 2 * Tc)
        ql = q[:, pos].clone().requires_grad_()
        kl, vl = k.clone().requires_grad_(), v.clone().requires_grad_()
        ol = F.packed_attention_sharded(ql, kl, vl, mask, shard)
        ol.backward(do[:, pos].contiguous())
        assert torch.equal(ol, of[:, pos]), f"rank {r}: forward differs from the full kernel"
        _close(ql.grad, qf.grad[:, pos], 1e-6, 0, f"rank {r} dQ")
        dk_sum += kl.grad.float()
        dv_sum += vl.grad.float()
    _close(dk_sum, kf.grad, 3e-2, 2e-2, "sum of partial dK")
    _close(dv_sum, vf.grad, 3e-2, 2e-2, "sum of partial dV")


def _attn_rows_fp64(q, k, v, doc, rows, scale):
    """Direct evaluation of the document-masked causal attention for a few query rows of one head (fp64 on the host):
    q [T, D], k / v [T, D] (bf16-rounded values as float64), doc [T] -> [len(rows), D]."""
    out = []
    for r in rows:
        allow = (doc[: r + 1] == doc[r]) & (doc[r] > 0)
        if not allow.any():
            out.append(torch.zeros(q.shape[1], dtype=torch.float64))
            continue
        s = (k[: r + 1][allow] @ q[r]) * scale
        p = torch.softmax(s, dim=0)
        out.append(p @ v[: r + 1][allow])
    return torch.stack(out)


def test_config_d_long_audio_context_parallel_segments():
    """BASELINE config D at full size: one row of T = 65536 holding two 30 000-token recordings + short fill, Qwen2-Audio
    head shape, cp = 4 with head/tail load balancing = 16384 query rows per rank in two 8192-row segments
    (`tn_attn_fwd_seg` / `tn_attn_bwd_seg`, global K/V + document ids).  Checked against
      * direct fp64 evaluation of the masked softmax on sampled rows (first / last rows of every segment, rows next
        to the document boundaries, pad rows), forward;
      * the size-independent properties: every rank's forward and dQ equal the rows of the FULL-sequence kernel, the
        four ranks' partial dK / dV sum to the full dK / dV, and the packed row equals each 30 000-token document
        run on its own (pack-vs-pad, tests/touchnet/utils/test_pack_loss.py's property), forward and backward."""
    F = _f()
    B, T, Nh, Nkv, D, cp = 1, 65536, 4, 4, 128, 4
    lens = [30000, 30000, 700, 1200, 900, 2000]                       # 64800 tokens, 736 pad positions
    doc = torch.zeros(B, T, dtype=torch.int64)
    t = 0
    for i, n in enumerate(lens):
        doc[0, t:t + n] = i + 1
        t += n
    g = torch.Generator().manual_seed(65536)
    q, k, v, do = [torch.randn(B, T, n, D, generator=g).bfloat16().to(DEV) for n in (Nh, Nkv, Nkv, Nh)]
    mask = F.build_packed_mask(doc.to(DEV))
    qf, kf, vf = [x.clone().requires_grad_() for x in (q, k, v)]
    of = F.packed_attention(qf, kf, vf, mask)
    of.backward(do)
    scale = D ** -0.5
    Tc = T // (2 * cp)                                                # 8192
    dk_sum, dv_sum = torch.zeros_like(k, dtype=torch.float32), torch.zeros_like(v, dtype=torch.float32)
    docs_cpu = doc[0]
    for r in range(cp):
        offs = (r * Tc, (2 * cp - 1 - r) * Tc)
        pos = torch.cat([torch.arange(o, o + Tc) for o in offs])
        shard = F.SeqShard(((0, Tc, offs[0]), (Tc, Tc, offs[1])), 2 * Tc)
        ql = q[:, pos].clone().requires_grad_()
        kl, vl = k.clone().requires_grad_(), v.clone().requires_grad_()
        ol = F.packed_attention_sharded(ql, kl, vl, mask, shard)
        ol.backward(do[:, pos].contiguous())
        assert torch.equal(ol, of[:, pos]), f"rank {r}: forward differs from the full-sequence kernel"
        _close(ql.grad, qf.grad[:, pos], 1e-6, 0, f"rank {r} dQ")
        dk_sum += kl.grad.float()
        dv_sum += vl.grad.float()
        # direct evaluation on sampled rows of this rank (head 1)
        rows = sorted({offs[0], offs[0] + 1, offs[0] + Tc - 1, offs[1], offs[1] + Tc // 2, offs[1] + Tc - 1}
                      | {x for x in (29999, 30000, 30001, 59999, 60000, 60699, 60700) if any(o <= x < o + Tc for o in offs)})
        want = _attn_rows_fp64(q[0, :, 1].double().cpu(), k[0, :, 1].double().cpu(), v[0, :, 1].double().cpu(),
                               docs_cpu, rows, scale)
        local = [int((pos == x).nonzero()[0, 0]) for x in rows]
        _close(ol[0, local, 1], want, 2e-2, 2e-2, f"rank {r}: sampled rows vs direct evaluation")
    _close(dk_sum, kf.grad, 6e-2, 3e-2, "sum of the four partial dK")
    _close(dv_sum, vf.grad, 6e-2, 3e-2, "sum of the four partial dV")
    # packed == each long document on its own
    for s_, e_ in ((0, 30000), (30000, 60000)):
        q1, k1, v1 = [x[:, s_:e_].clone().requires_grad_() for x in (q, k, v)]
        o1 = F.packed_attention(q1, k1, v1, F.causal_mask(1, e_ - s_, DEV))
        o1.backward(do[:, s_:e_].contiguous())
        _close(of[:, s_:e_], o1, 1e-2, 1e-2, f"O of document [{s_}, {e_})")
        _close(qf.grad[:, s_:e_], q1.grad, 2e-2, 2e-2, "dQ")
        _close(kf.grad[:, s_:e_], k1.grad, 6e-2, 3e-2, "dK")
        _close(vf.grad[:, s_:e_], v1.grad, 6e-2, 3e-2, "dV")
    pad = docs_cpu == 0
    assert float(of[0][pad].float().abs().max()) == 0.0 and float(kf.grad[0][pad].float().abs().max()) == 0.0


def test_config_d_local_remote_split_with_lse_merge_equals_the_single_kernel():
    """Context parallel at config-D size (T = 65536, cp = 4, two 30 000-token recordings): the attention of a rank's
    16384 rows over its OWN two chunks + over the six other chunks, merged by tn_attn_merge, against the single
    tn_attn_fwd_seg over the global K/V — equal to bf16 rounding of the merge (the partial results are rounded to bf16
    before they are combined) — and the gradients of the split node (ONE tn_attn_bwd_seg with the merged O / LSE)
    against the unsplit node.  Also: a rank whose rows see NO key in their own chunks for one document part, pad rows,
    and the `wait` callback sitting exactly between the two parts."""
    F = _f()
    from touchnet_amd import library as Lb
    B, T, Nh, Nkv, D, cp = 1, 65536, 4, 4, 128, 4
    lens = [30000, 30000, 700, 1200, 900, 2000]
    doc = torch.zeros(B, T, dtype=torch.int64)
    t = 0
    for i, n in enumerate(lens):
        doc[0, t:t + n] = i + 1
        t += n
    g = torch.Generator().manual_seed(4 * 65536)
    q, k, v, do = [torch.randn(B, T, n, D, generator=g).bfloat16().to(DEV) for n in (Nh, Nkv, Nkv, Nh)]
    mask = F.build_packed_mask(doc.to(DEV))
    Tc = T // (2 * cp)
    for r in range(cp):
        mine = [r, 2 * cp - 1 - r]
        remote = [c for c in range(2 * cp) if c not in mine and c < max(mine)]      # (causal: later chunks see nothing)
        offs = (mine[0] * Tc, mine[1] * Tc)
        pos = torch.cat([torch.arange(o, o + Tc) for o in offs])
        shard = F.SeqShard(((0, Tc, offs[0]), (Tc, Tc, offs[1])), 2 * Tc)
        dol = do[:, pos].contiguous()
        ql, kl, vl = q[:, pos].clone().requires_grad_(), k.clone().requires_grad_(), v.clone().requires_grad_()
        ref = F.packed_attention_sharded(ql, kl, vl, mask, shard)
        ref.backward(dol)
        qs, ks, vs = q[:, pos].clone().requires_grad_(), k.clone().requires_grad_(), v.clone().requires_grad_()
        calls = []
        out = F.packed_attention_sharded_split(qs, ks, vs, mask, shard, Tc, mine, remote, wait=lambda: calls.append(1))
        out.backward(dol)
        assert calls == [1]
        _close(out, ref, 2 ** -7, 2 ** -7, f"rank {r}: merged forward vs single kernel")
        _close(qs.grad, ql.grad, 3e-2, 3e-2, f"rank {r}: dQ")
        _close(ks.grad, kl.grad, 6e-2, 3e-2, f"rank {r}: dK partial")
        _close(vs.grad, vl.grad, 6e-2, 3e-2, f"rank {r}: dV partial")
        # the two parts on their own: rows whose document lies entirely in remote chunks have an EMPTY local part
        segs = shard.flat()
        bits = lambda cs: sum(1 << c for c in cs)
        o_a, l_a = Lb.attn_fwd_seg_chunks(qs.detach(), k, v, mask.doc, mask.meta, D ** -0.5, segs, 2 * Tc, Tc, bits(mine))
        o_b, l_b = Lb.attn_fwd_seg_chunks(qs.detach(), k, v, mask.doc, mask.meta, D ** -0.5, segs, 2 * Tc, Tc, bits(remote))
        assert bool(torch.isinf(l_b[0, :, :Tc]).all()) == (r == 0)                  # chunk 0 has nothing before it
        empty_a = torch.isinf(l_a)
        assert float(o_a.transpose(1, 2)[empty_a].float().abs().max() if empty_a.any() else 0.0) == 0.0
        pad = (doc[0, pos] == 0).to(DEV)
        assert float(out[0][pad].float().abs().max() if pad.any() else 0.0) == 0.0


# ------------------------------------------------------------------------------------ linear layers
@pytest.mark.parametrize("R,C", [(8, 8), (256, 64), (264, 72), (1000 * 8, 1280), (16384, 4096)])
def test_transpose_bf16_bit_exact(R, C):
    """tn_transpose_bf16 is a byte permutation: bit-exact against torch, incl. strided source / destination."""
    F = _f()
    x = torch.randn(R, C + 8, device=DEV).to(torch.bfloat16)[:, :C]          # row stride C + 8
    out = torch.full((C, R + 16), 7.0, dtype=torch.bfloat16, device=DEV)
    F.transpose_2d(x, out=out[:, :R])
    assert torch.equal(out[:, :R], x.t())
    assert torch.equal(out[:, R:], torch.full((C, 16), 7.0, dtype=torch.bfloat16, device=DEV))   # nothing spilled
    assert torch.equal(F.transpose_2d(x.contiguous()), x.t().contiguous())
    with pytest.raises(RuntimeError):
        F.transpose_2d(x[:, :C - 1])                                           # not a multiple of 8 -> -22, loud


@pytest.mark.parametrize("M,H,I", [(256, 128, 512), (1000, 256, 688), (4104, 512, 1376)])
def test_swiglu_mlp_fused_node(M, H, I):
    """The fused MLP node: (1) its SwiGLU kernels are bit-identical to the plain ones and their transposed outputs
    are exact transposes; (2) output and all four gradients against autograd over nn.Linear / silu in fp32 on the same
    bf16-rounded inputs."""
    F = _f()
    from touchnet_amd import _C
    g = torch.Generator(device="cpu").manual_seed(M)
    gate = torch.randn(M, I, generator=g).bfloat16().to(DEV)
    up = torch.randn(M, I, generator=g).bfloat16().to(DEV)
    dact = torch.randn(M, I, generator=g).bfloat16().to(DEV)
    act, act_t = torch.empty_like(gate), torch.empty(I, M, dtype=torch.bfloat16, device=DEV)
    _C.check(_C.lib().tn_swiglu_fwd_t(_C.ptr(gate), _C.ptr(up), _C.ptr(act), _C.ptr(act_t), M, I, _C.stream()), "fwd_t")
    gg, uu = gate.clone().requires_grad_(), up.clone().requires_grad_()
    ref = F.swiglu(gg, uu)
    ref.backward(dact)
    assert torch.equal(act, ref) and torch.equal(act_t, ref.t())
    dg, du = torch.empty_like(gate), torch.empty_like(up)
    dgu_t = torch.empty(2 * I, M, dtype=torch.bfloat16, device=DEV)
    _C.check(_C.lib().tn_swiglu_bwd_t(_C.ptr(dact), _C.ptr(gate), _C.ptr(up), _C.ptr(dg), _C.ptr(du), _C.ptr(dgu_t), M, I,
                                      _C.stream()), "bwd_t")
    assert torch.equal(dg, gg.grad) and torch.equal(du, uu.grad)
    assert torch.equal(dgu_t[:I], dg.t()) and torch.equal(dgu_t[I:], du.t())

    x = (torch.randn(2, M // 2, H, generator=g) * 0.5).bfloat16()
    ws = [(torch.randn(n, k, generator=g) * k ** -0.5).bfloat16() for n, k in ((I, H), (I, H), (H, I))]
    dy = torch.randn(2, M // 2, H, generator=g).bfloat16()

    def run(dev, dt, fn):
        xx = x.to(dev, dt).requires_grad_()
        ww = [w.to(dev, dt).requires_grad_() for w in ws]
        y = fn(xx, *ww)
        y.backward(dy.to(dev, dt))
        return [y, xx.grad] + [w.grad for w in ww]

    lin = torch.nn.functional.linear
    want = run("cpu", torch.float32, lambda xx, a, b, c: lin(torch.nn.functional.silu(lin(xx, a)) * lin(xx, b), c))
    got = run(DEV, torch.bfloat16, F.swiglu_mlp)
    for name, a, b in zip(("y", "dx", "dWgate", "dWup", "dWdown"), got, want):
        assert a.shape == b.shape
        torch.testing.assert_close(a.float().cpu(), b, rtol=3e-2, atol=3e-2 * float(b.abs().max()), msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("R,C", [(7, 8), (64, 1280), (30000, 1280), (4097, 5120), (16384, 4096)])
def test_column_sum_bias_gradient(R, C):
    """tn_colsum_bf16 vs an fp64 column sum of the same bf16 values: fp32 accumulation, one bf16 rounding; also on a
    column slice of a wider matrix (row stride > cols) and deterministic."""
    F = _f()
    x = torch.randn(R, C + 16, device=DEV).to(torch.bfloat16)
    for view in (x[:, :C], x[:, 8:8 + C].contiguous()):
        got = F.column_sum(view)
        ref = view.double().sum(0)
        assert got.dtype == torch.bfloat16 and got.shape == (C,)
        tol = 2.0 ** -8 * ref.abs().clamp_min(1.0) + 1e-3 * math.sqrt(R)     # bf16 ulp of the result + fp32 sum error
        assert ((got.double() - ref).abs() <= tol).all(), float((got.double() - ref).abs().max())
        assert torch.equal(got, F.column_sum(view))


@pytest.mark.parametrize("M,K,Ns,bias,wtn", [(512, 256, (256, 64, 64), True, "tn"), (2048, 1024, (2816, 2816), False, "tn"),
                                             (384, 128, (128,), False, "tn"), (384, 2816, (1024,), False, "nt"),
                                             (1000, 1280, (1280, 1280, 1280), True, "nt_fused"),
                                             (100, 64, (64, 32), True, "tn")])
def test_linear_group_matches_autograd(M, K, Ns, bias, wtn):
    """Forward = nn.Linear; backward (transposed-operand weight-gradient GEMM over the whole group, addmm-accumulated
    input gradient) against autograd's own nn.Linear backward in fp32 on the same bf16-rounded inputs.
    M = 100 is not a multiple of 8: the HIP transposes do not apply and the torch path must give the same."""
    F = _f()
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.randn(2, M // 2, K, generator=g) * 0.5).to(torch.bfloat16)
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16) for N in Ns]
    bs = [(torch.randn(N, generator=g) * 0.1).to(torch.bfloat16) if bias else None for N in Ns]
    dys = [torch.randn(2, M // 2, N, generator=g).to(torch.bfloat16) for N in Ns]

    def run(dev, dt, fn):
        xx = x.to(dev, dt).requires_grad_()
        ww = [w.to(dev, dt).requires_grad_() for w in ws]
        bb = [None if b is None else b.to(dev, dt).requires_grad_() for b in bs]
        ys = fn(xx, list(zip(ww, bb)))
        torch.autograd.backward(ys, [d.to(dev, dt) for d in dys])
        return ys, xx.grad, [w.grad for w in ww], [None if b is None else b.grad for b in bb]

    ref = run("cpu", torch.float32, lambda xx, layers: [torch.nn.functional.linear(xx, w, b) for w, b in layers])
    got = run(DEV, torch.bfloat16, lambda xx, layers: F.linear_group(xx, layers, wgrad=wtn, dgrad_tn=wtn != "nt_fused"))
    for a, b in zip(got[0], ref[0]):
        torch.testing.assert_close(a.float().cpu(), b, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(got[1].float().cpu(), ref[1], rtol=2e-2, atol=2e-2 * float(ref[1].abs().max()))
    for a, b in zip(got[2], ref[2]):
        assert a.shape == b.shape
        torch.testing.assert_close(a.float().cpu(), b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
    for a, b in zip(got[3], ref[3]):
        if b is not None:
            torch.testing.assert_close(a.float().cpu(), b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))


def test_linear_layers_on_the_hand_written_gemm_match_the_library(monkeypatch):
    """TN_LINEAR_GEMM=own (`bench.py --linear-gemm own`): every product of linear_group and of the fused SwiGLU MLP node —
    forward, input gradient, weight gradient — runs on csrc/gemm.hip in its native operand mode instead of hipBLASLt on
    transposed copies.  Same bf16 inputs, fp32 accumulation in both: outputs and all gradients agree to bf16 rounding of
    differently ordered sums, the kernel really is the one that ran, and no transpose pass is left."""
    F = _f()
    from touchnet_amd import _C
    g = torch.Generator(device="cpu").manual_seed(5)
    M, H, I = 1024, 256, 512
    x = (torch.randn(2, M // 2, H, generator=g) * 0.5).bfloat16().to(DEV)
    ws = [(torch.randn(n, k, generator=g) * k ** -0.5).bfloat16().to(DEV) for n, k in ((I, H), (I, H), (H, I))]
    qkv = [((torch.randn(n, H, generator=g) * H ** -0.5).bfloat16().to(DEV), (torch.randn(n, generator=g) * 0.1).bfloat16().to(DEV))
           for n in (256, 128, 128)]
    dy = torch.randn(2, M // 2, H, generator=g).bfloat16().to(DEV)
    dq = [torch.randn(2, M // 2, n, generator=g).bfloat16().to(DEV) for n in (256, 128, 128)]

    def run(mode):
        monkeypatch.setattr(F, "LINEAR_GEMM", mode)
        monkeypatch.setattr(F, "_OWN_MIN_TILES", 1)        # (test sizes have a handful of tiles)
        calls = []
        monkeypatch.setattr(F, "gemm", lambda *a, **k: (calls.append((len(a[0]), a[1:], tuple(k))), _orig(*a, **k))[1])
        for name in fused:      # the round-5 launches: SwiGLU epilogues (fwd, bwd) and the grouped weight gradients
            monkeypatch.setattr(F, name, (lambda o, nm: lambda *a, **k: (calls.append((nm, (), ())), o(*a, **k))[1])(
                fused[name], name))
        tr = []
        monkeypatch.setattr(F, "transpose_2d", lambda *a, **k: (tr.append(1), _orig_t(*a, **k))[1])
        xx = x.clone().requires_grad_()
        ww = [w.clone().requires_grad_() for w in ws]
        y = F.swiglu_mlp(xx, *ww)
        y.backward(dy)
        x2 = x.clone().requires_grad_()
        lw = [(w.clone().requires_grad_(), b.clone().requires_grad_()) for w, b in qkv]
        outs = F.linear_group(x2, lw)
        torch.autograd.backward(outs, dq)
        res = [y, xx.grad] + [w.grad for w in ww] + list(outs) + [x2.grad] + [w.grad for w, _ in lw] + [b.grad for _, b in lw]
        return [r.float() for r in res], calls, len(tr)

    _orig, _orig_t = F.gemm, F.transpose_2d
    fused = {n: getattr(F, n) for n in ("gemm_swiglu_fwd", "gemm_swiglu_bwd", "gemm_grouped_wgrad")}
    lib, c_lib, t_lib = run("lib")
    own, c_own, t_own = run("own")
    # MLP (round 5): gate + up + SwiGLU as ONE launch, down; d(act) + SwiGLU backward as ONE launch, dX (one launch over
    # the gate / up pair), the three weight gradients as ONE grouped launch = 5 launches (round 4: 8 + 2 SwiGLU kernels);
    # group: 3 forward, dX (ONE launch with three segments), 3 dW.  And not a single transposed copy.
    assert len(c_lib) == 0 and t_lib > 0
    assert len(c_own) == 5 + 7 and t_own == 0, (len(c_own), t_own)
    assert sorted(str(n) for n, _, _ in c_own) == sorted(["1"] * 7 + ["2", "3"] + list(fused))
    for i, (a, b) in enumerate(zip(own, lib)):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1.6e-2 * scale + 1e-6, (i, float((a - b).abs().max()), scale)


# ------------------------------------------------------------------------------------ BEST-RQ tokenizer (§8f-4)
@pytest.mark.parametrize("case", ["recipe", "default", "small"])
def test_bestrq_tokenize_matches_reference_codes(golden, case):
    """tn_bestrq_tokenize through the product BestRQTokenizer (tables drawn like the reference's) against the codes
    the reference produced for the same features: identical wherever the reference's own best/second-best margin
    exceeds 1e-6 (fp32 round-off of a distance ~1), and at most 1 % near-tie differences overall."""
    import types
    from touchnet_amd.tokenizer import BestRQTokenizer
    g = golden("bestrq.npz")
    V, Fdim, E, seed = [int(v) for v in g[f"{case}/cfg"]]
    cfg = types.SimpleNamespace(tokenizer_bestrq_vocab_size=V, tokenizer_bestrq_input_size=Fdim,
                                tokenizer_bestrq_emb_size=E, tokenizer_bestrq_init_seed=seed,
                                tokenizer_bestrq_init_method="default", tokenizer_type="BestRQTokenizer")
    tok = BestRQTokenizer(cfg, device=DEV)
    codes = tok.tokenize(torch.from_numpy(g[f"{case}/feat"]).to(DEV))
    assert codes.dtype == torch.int64 and codes.is_cuda
    big = V > 1024
    np.testing.assert_array_equal(tok._codebook.cpu().numpy()[::64] if big else tok._codebook.cpu().numpy(),
                                  g[f"{case}/codebook"])
    bad = codes.cpu().numpy() != g[f"{case}/codes"]
    assert not (bad & (g[f"{case}/margin"] > 1e-6)).any(), g[f"{case}/margin"][bad]
    assert bad.mean() <= 0.01
    with pytest.raises(RuntimeError):
        tok.tokenize(torch.from_numpy(g[f"{case}/feat"]))          # CPU tensor: refused, no host fallback


def test_bestrq_tokenize_full_size_properties():
    """T = 131072 frames (the recipe's row length), V = 8192: against the oracle on a 2048-frame sample, plus
    size-independent properties: codes in range, invariance to positive scaling of a frame (normalisation), ragged
    tail (T not a multiple of the 32-frame tile) equals the prefix of the padded call."""
    from oracle import tokenizer as otok
    F = _f()
    q, c = otok.bestrq_tables(8192, 512, 16, 2026)
    qd, cd = torch.from_numpy(q).to(DEV), torch.from_numpy(c).to(DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    feat = torch.randn(131072, 512, generator=g)
    fd = feat.to(DEV)
    codes = F.bestrq_tokenize(fd, qd, cd)
    assert int(codes.min()) >= 0 and int(codes.max()) < 8192
    idx = torch.randperm(131072, generator=g)[:2048]
    ref = otok.bestrq_tokenize(feat[idx].numpy(), q, c)
    assert (codes[idx.to(DEV)].cpu().numpy() != ref).mean() <= 0.002       # near-ties only
    scale = (torch.rand(4096, 1, generator=g) * 9 + 0.5).to(DEV)
    assert (F.bestrq_tokenize(fd[:4096] * scale, qd, cd) != codes[:4096]).float().mean() <= 0.002
    assert torch.equal(F.bestrq_tokenize(fd[:1000 + 13], qd, cd), codes[:1013])
    assert F.bestrq_tokenize(fd[:0], qd, cd).numel() == 0


# ------------------------------------------------------------------------------------ int16 PCM on the device (§8f-3)
def test_pcm16_to_float_exact_and_frontend_equivalence(golden):
    """tn_pcm16_to_f32 == numpy's astype(float32) / 32768 for EVERY int16 value (bit exact), on ragged / unaligned
    slices, and the log-mel stage fed int16 samples equals the stage fed the reference's float waveform."""
    import types
    F = _f()
    allv = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16)
    want = torch.from_numpy(allv.numpy().astype(np.float32) / 32768.0)
    assert torch.equal(F.pcm16_to_float(allv.to(DEV)).cpu(), want)
    for off, n in ((1, 17), (3, 4099), (0, 8), (5, 0)):
        assert torch.equal(F.pcm16_to_float(allv.to(DEV)[off:off + n]).cpu(), want[off:off + n])
    with pytest.raises(RuntimeError):
        F.pcm16_to_float(allv)                                   # CPU tensor: refused
    from touchnet_amd.data import functions as stages
    pcm = torch.from_numpy(golden("logmel.npz")["wav0/pcm"].astype(np.int16))[None]
    cfg = types.SimpleNamespace(audiofeat_padding=0, audiofeat_n_fft=400, audiofeat_hop_length=160,
                                audiofeat_num_mel_bins=128)
    a = next(stages.audio_compute_log_mel_spectrogram(iter([{"sample_rate": 16000, "waveform": pcm.clone()}]), cfg))
    b = next(stages.audio_compute_log_mel_spectrogram(
        iter([{"sample_rate": 16000, "waveform": pcm.to(torch.float32) / 32768.0}]), cfg))
    assert torch.equal(a["audiofeat"], b["audiofeat"])


@pytest.mark.parametrize("D,Nh,Nkv", [(128, 2, 2), (64, 4, 2)])
def test_attention_long_sequence_multi_chunk_tile_list(D, Nh, Nkv):
    """T = 32768 (512 KV tiles): the forward walks its LDS tile list in several chunks (192 tiles each in the default
    schedule).  One 30000-token document followed by short ones; checked on query slices that sit before, on and after
    the chunk boundaries against a direct fp32 evaluation of the same bf16 inputs, plus the packed == per-document
    property on a short trailing document, run-to-run bit identity and the dQ rows of the same slices."""
    F = _f()
    B, T = 1, 32768
    doc = torch.zeros(B, T, dtype=torch.int32)
    doc[0, :30000] = 1
    doc[0, 30000:31000] = 2
    doc[0, 31000:32700] = 3                     # 68 pad positions at the end
    g = torch.Generator().manual_seed(21)
    q, k, v, do = [torch.randn(B, T, n, D, generator=g).bfloat16().to(DEV) for n in (Nh, Nkv, Nkv, Nh)]
    mask = F.build_packed_mask(doc.to(DEV))
    qg, kg, vg = [t.clone().requires_grad_() for t in (q, k, v)]
    out = F.packed_attention(qg, kg, vg, mask)
    out.backward(do)
    assert torch.equal(out, F.packed_attention(q, k, v, mask))
    G = Nh // Nkv
    kf, vf = k.float().repeat_interleave(G, dim=2), v.float().repeat_interleave(G, dim=2)
    ids = doc[0].to(DEV)
    for s in (0, 12224, 12288 + 64, 24576 - 32, 29900, 30950, 32640):       # around multiples of 192 tiles = 12288 tokens
        rows = torch.arange(s, min(s + 96, T), device=DEV)
        sc = torch.einsum("rhd,thd->hrt", q[0, rows].float(), kf[0]) * D ** -0.5
        allow = ((ids[None, :] == ids[rows][:, None]) & (ids[rows][:, None] > 0)
                 & (torch.arange(T, device=DEV)[None, :] <= rows[:, None]))
        sc = sc.masked_fill(~allow[None], float("-inf"))
        p = torch.softmax(sc, dim=-1)
        p = torch.nan_to_num(p, nan=0.0)                                      # fully masked (pad) rows -> 0
        ref = torch.einsum("hrt,thd->rhd", p, vf[0])
        _close(out[0, rows], ref, 1e-2, 1e-2, f"O rows {s}..")
        dp = torch.einsum("rhd,thd->hrt", do[0, rows].float(), vf[0])
        delta = (do[0, rows].float() * ref).sum(-1).transpose(0, 1)          # [h, r]
        ds = p * (dp - delta[..., None])
        dq_ref = torch.einsum("hrt,thd->rhd", ds, kf[0]) * D ** -0.5
        _close(qg.grad[0, rows], dq_ref, 3e-2, 3e-2, f"dQ rows {s}..")
    # dK / dV of KV rows whose workgroup walks MANY query stages (the dK/dV kernel compacts 256 candidate stages at a time
    # into its LDS list: the first KV rows of the long document meet 469 q tiles x G heads = several list chunks).
    # Reference: fp32 softmax statistics of every row of the long document, computed chunk-wise.
    L = 30000
    scale = D ** -0.5
    qf, dof = q[0, :L].float(), do[0, :L].float()
    lse = torch.empty(Nh, L, device=DEV)
    dlt = torch.empty(Nh, L, device=DEV)
    pos = torch.arange(L, device=DEV)
    for s in range(0, L, 2048):
        e = min(s + 2048, L)
        sc = torch.einsum("rhd,thd->hrt", qf[s:e], kf[0, :L]) * scale
        sc = sc.masked_fill((pos[None, :] > pos[s:e, None])[None], float("-inf"))
        lse[:, s:e] = torch.logsumexp(sc, dim=-1)
        o_ref = torch.einsum("hrt,thd->rhd", torch.softmax(sc, dim=-1), vf[0, :L])
        dlt[:, s:e] = (dof[s:e] * o_ref).sum(-1).transpose(0, 1)
        del sc, o_ref
    for ks in (0, 16384 - 32, 29936):
        kr = torch.arange(ks, ks + 64, device=DEV)
        sc = torch.einsum("rhd,thd->hrt", qf, kf[0, kr]) * scale                      # [Nh, L, 64]
        p = torch.exp(sc - lse[..., None]).masked_fill((kr[None, :] > pos[:, None])[None], 0.0)
        dv_ref = torch.einsum("hrt,rhd->thd", p, dof)                                   # [64, Nh, D]
        ds = p * (torch.einsum("rhd,thd->hrt", dof, vf[0, kr]) - dlt[..., None])
        dk_ref = torch.einsum("hrt,rhd->thd", ds, qf) * scale
        fold = lambda t: t.view(64, Nkv, G, D).sum(2)                                   # GQA: sum over the group's heads
        _close(vg.grad[0, kr], fold(dv_ref), 3e-2, 3e-2, f"dV rows {ks}..")
        _close(kg.grad[0, kr], fold(dk_ref), 3e-2, 3e-2, f"dK rows {ks}..")
        del sc, p, ds
    q1, k1, v1 = [t[:, 30000:31000].clone().requires_grad_() for t in (q, k, v)]
    o1 = F.packed_attention(q1, k1, v1, F.causal_mask(1, 1000, DEV))
    _close(out[:, 30000:31000], o1, 1e-2, 1e-2, "short document after the long one")
    assert float(out[0, 32700:].float().abs().max()) == 0.0


# ---------------------------------------------------------------------------------------- hand-written MFMA GEMM
@pytest.mark.parametrize("M,N,K,bias,acc,ct", [
    (256, 256, 128, False, False, False),      # one tile, two stages
    (512, 768, 256, True, False, False),       # bias, several tiles
    (300, 264, 384, False, False, False),      # ragged M and N (zero-filled DMA rows, guarded stores)
    (1000, 1288, 512, True, True, False),      # accumulate into C (group input gradients), ragged
    (512, 512, 1024, False, False, True),      # transposed copy
    (776, 1032, 256, True, False, True),       # transposed copy, ragged, bias
    (2048, 4096, 4096, False, False, False),   # decoder-block shape (o_proj / q_proj), XCD tile map with 128 tiles
    (1024, 11008, 4096, False, False, False),  # gate / up: 43 column tiles (odd count through the bijective map)
])
def test_gemm_tn_matches_fp32_reference(M, N, K, bias, acc, ct):
    """tn_gemm_bf16_tn vs an fp32 torch.mm of the same bf16-rounded operands (asymmetric random data, so a swapped or
    transposed tile cannot pass); tolerance = half a bf16 ulp of the result + fp32 summation-order noise."""
    F = _f()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)
    bv = torch.randn(N, generator=g).to(torch.bfloat16) if bias else None
    c0 = torch.randn(M, N, generator=g).to(torch.bfloat16) if acc else None
    ref = a.double() @ b.double().t()
    if bias:
        ref = ref + bv.double()
    if acc:
        ref = ref + c0.double()
    ad, bd = a.to(DEV), b.to(DEV)
    out = c0.to(DEV).clone() if acc else None
    out_t = torch.full((N, M), float("nan"), dtype=torch.bfloat16, device=DEV) if ct else None
    got = F.gemm_tn(ad, bd, bias=bv.to(DEV) if bias else None, out=out, accumulate=acc, out_t=out_t)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    _close(got, ref, atol=scale * 2 ** -8, rtol=2 ** -7, what=f"gemm_tn {M}x{N}x{K}")
    if ct:
        assert torch.equal(out_t.cpu(), got.cpu().t()), "transposed copy differs from C^T"


@pytest.mark.parametrize("mode", ["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("M,N,K", [
    (256, 256, 64),          # one tile, one stage: prologue / epilogue only
    (520, 264, 192),         # ragged rows and columns
    (1000, 776, 1088),       # 17 stages: the ring wraps more than three times
    (4360, 4104, 128),       # 306 tiles > 256 CUs: persistent workgroups walk two tiles (park in the freed slots)
    (1280, 520, 3000),       # depth not a multiple of 64: weight-gradient mode only (zero-filled tail stage)
])
def test_gemm_operand_modes_match_fp64_reference(mode, M, N, K):
    """The three products of a linear layer in their native layouts (functional.gemm): forward x W^T, input gradient
    dY W (W contraction-major, read with ds_read_b64_tr_b16), weight gradient dY^T x (both contraction-major).
    Asymmetric random data: a swapped operand, a transposed tile or a wrong contraction slot cannot pass."""
    F = _f()
    if K % 64 and mode != "wgrad":
        from touchnet_amd import _C
        with pytest.raises(_C.KernelError):
            F.gemm([(torch.zeros(M, K, dtype=torch.bfloat16, device=DEV),
                     torch.zeros(*((N, K) if mode == "fwd" else (K, N)), dtype=torch.bfloat16, device=DEV))],
                   False, mode == "dgrad")
        return
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).to(torch.bfloat16)
    if mode == "fwd":
        a, b = r(M, K), r(N, K)
        ref, ak, bk = a.double() @ b.double().t(), False, False
    elif mode == "dgrad":
        a, b = r(M, K), r(K, N)
        ref, ak, bk = a.double() @ b.double(), False, True
    else:
        a, b = r(K, M), r(K, N)
        ref, ak, bk = a.double().t() @ b.double(), True, True
    got = F.gemm([(a.to(DEV), b.to(DEV))], ak, bk)
    _close(got, ref, atol=float(ref.abs().max()) * 2 ** -8, rtol=2 ** -7, what=f"gemm {mode} {M}x{N}x{K}")


@pytest.mark.parametrize("mode,M,N,K", [
    ("wgrad", 1280, 1280, 30000),     # the audio tower's q/k/v/o weight gradients: 25 tiles, ragged depth, 10 parts
    ("wgrad", 520, 264, 2500),        # ragged everything; the last part is cut by the descriptor
    ("wgrad", 1280, 5120, 30000),     # 100 tiles: 2 parts
    ("fwd", 1024, 512, 4096),         # contraction-contiguous operands: 64 stages split evenly
    ("dgrad", 1000, 776, 2048),
    ("wgrad", 11008, 4096, 2048),     # TAIL split: 688 tiles = 2 whole rounds unsplit + 176 tiles in 4 parts
    ("fwd", 4360, 4104, 1024),        # 306 tiles = 1 round + 50 ragged-edge tiles in 2 parts
    ("dgrad", 4360, 4104, 1024),
])
def test_gemm_split_k_matches_fp64_reference(mode, M, N, K, monkeypatch):
    """Outputs of few tiles with a deep contraction run as tiles x parts units (tn_gemm_bf16_splitk: fp32 partial sums
    through a workspace, summed with bias / accumulate by a second kernel) — same result as the unsplit kernel up to the
    summation order, equal to fp64 rounded once."""
    F = _f()
    parts = F.split_k(M, N, K, mode == "wgrad", mode != "fwd")
    tail = parts == 1          # many tiles: only the last partial round is split (tn_gemm_bf16_splitk tail_only = 1, direct call)
    if tail:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        assert tiles > 256 and tiles % 256
        parts = 4 if mode == "wgrad" else 2
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).to(torch.bfloat16)
    if mode == "fwd":
        a, b = r(M, K), r(N, K)
        ref, ak, bk = a.double() @ b.double().t(), False, False
    elif mode == "dgrad":
        a, b = r(M, K), r(K, N)
        ref, ak, bk = a.double() @ b.double(), False, True
    else:
        a, b = r(K, M), r(K, N)
        ref, ak, bk = a.double().t() @ b.double(), True, True
    a, b = a.to(DEV), b.to(DEV)

    def product(bias=None, out=None, accumulate=False):
        if not tail:
            return F.gemm([(a, b)], ak, bk, bias=bias, out=out, accumulate=accumulate)
        from touchnet_amd import _C
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV) if out is None else out
        ws = torch.empty(parts * (tiles % 256) * 65536, dtype=torch.float32, device=DEV)
        ptr = lambda t: None if t is None else t.data_ptr()
        _C.check(_C.lib().tn_gemm_bf16_splitk(ptr(a), ptr(b), a.stride(0), b.stride(0), K, int(ak), int(bk), ptr(out), ptr(bias),
                                              M, N, out.stride(0), int(accumulate), parts, 1, ptr(ws), ws.numel() * 4,
                                              torch.cuda.current_stream().cuda_stream), "tn_gemm_bf16_splitk")
        return out

    got = product()
    tol = dict(atol=float(ref.abs().max()) * 2 ** -8, rtol=2 ** -7)
    _close(got, ref, **tol, what=f"split-k gemm {mode} {M}x{N}x{K} / {parts}")
    monkeypatch.setattr(F, "SPLIT_K", False)
    whole = F.gemm([(a, b)], ak, bk)
    assert float((whole.float() - got.float()).abs().max()) <= float(ref.abs().max()) * 2 ** -7
    monkeypatch.setattr(F, "SPLIT_K", True)
    bias = r(N).to(DEV)
    base = r(M, N).to(DEV)
    out = base.clone()
    product(bias=bias, out=out, accumulate=True)
    _close(out, ref + bias.double().cpu() + base.double().cpu(), **tol, what="split-k bias + accumulate")
    assert torch.equal(product(), got)                      # deterministic


@pytest.mark.parametrize("mode,M,N,K", [("fwd", 4360, 4104, 128), ("dgrad", 1000, 776, 1088), ("wgrad", 1280, 1280, 30000),
                                        ("wgrad", 1280, 520, 3000)])
def test_gemm_one_workgroup_per_tile_launch(mode, M, N, K, monkeypatch):
    """TN_GEMM_PERSIST=0 (what the Trainer selects when collectives run beside the compute): grid = tiles (x split-K parts),
    every workgroup handles exactly one unit — same results as the persistent launch, bit for bit."""
    F = _f()
    g = torch.Generator().manual_seed(M + N + K)
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    a, b = (r(M, K), r(N, K)) if mode == "fwd" else (r(M, K), r(K, N)) if mode == "dgrad" else (r(K, M), r(K, N))
    ak, bk = mode == "wgrad", mode != "fwd"
    monkeypatch.delenv("TN_GEMM_PERSIST", raising=False)
    persistent = F.gemm([(a, b)], ak, bk)
    monkeypatch.setenv("TN_GEMM_PERSIST", "0")
    per_tile = F.gemm([(a, b)], ak, bk)
    assert torch.equal(persistent, per_tile)


@pytest.mark.parametrize("M,N,K", [(4096, 4104, 2048), (1280, 1280, 30000), (520, 264, 2500)])
def test_gemm_weight_gradient_with_fp32_output(M, N, K):
    """tn_gemm_bf16_wgrad_f32: dW = dY^T x written as fp32 (plain launch and split-K), overwrite and accumulate — the
    accumulators unrounded: equal to the fp64 product to fp32 accuracy, far below a bf16 ulp."""
    F = _f()
    g = torch.Generator().manual_seed(M + N + K)
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).to(torch.bfloat16)
    a, b = r(K, M), r(K, N)
    ref = a.double().t() @ b.double()
    out = torch.full((M, N), 7.0, dtype=torch.float32, device=DEV)
    F.gemm([(a.to(DEV), b.to(DEV))], True, True, out=out)
    scale = float(ref.abs().max())
    assert float((out.double().cpu() - ref).abs().max()) < scale * 2e-6 * (K ** 0.5), "fp32 weight gradient"
    F.gemm([(a.to(DEV), b.to(DEV))], True, True, out=out, accumulate=True)
    assert float((out.double().cpu() - 2 * ref).abs().max()) < scale * 4e-6 * (K ** 0.5), "fp32 accumulate"
    # into a view of a larger flat buffer (what the data-parallel engine hands out)
    flat = torch.zeros(128 + M * N + 64, dtype=torch.float32, device=DEV)
    view = flat[128:128 + M * N].view(M, N)
    F.gemm([(a.to(DEV), b.to(DEV))], True, True, out=view)
    assert torch.equal(view, (out - view)) or float((view.double().cpu() - ref).abs().max()) < scale * 2e-6 * (K ** 0.5)
    assert float(flat[:128].abs().max()) == 0.0 and float(flat[128 + M * N:].abs().max()) == 0.0
    with pytest.raises(Exception):
        F.gemm([(a.t().contiguous().to(DEV), b.t().contiguous().to(DEV))], False, False, out=out)     # fp32: wgrad mode only


def test_gemm_segments_accumulate_in_fp32_and_reject_bad_shapes():
    """dX = dQ Wq + dK Wk + dV Wv as ONE launch (three segments of different depth and row pitch) and dW over two token
    ranges: equal to the fp64 sum rounded once — better than three bf16 round trips through C."""
    F = _f()
    from touchnet_amd import _C
    g = torch.Generator().manual_seed(77)
    r = lambda *sh: (torch.rand(*sh, generator=g) * 2 - 1).to(torch.bfloat16).to(DEV)
    M, N = 1032, 520
    segs, ref = [], 0
    for K in (256, 64, 128):
        a, b = r(M, K + 64)[:, :K], r(K, N)                  # (row pitch K + 64)
        segs.append((a, b))
        ref = ref + a.double().cpu() @ b.double().cpu()
    got = F.gemm(segs, False, True)
    _close(got, ref, atol=float(ref.abs().max()) * 2 ** -8, rtol=2 ** -7, what="3-segment input gradient")
    chained = None
    for a, b in segs:                                        # the same sum with a bf16 rounding behind every term
        chained = F.gemm([(a, b)], False, True, out=chained, accumulate=chained is not None)
    e1 = float((got.double().cpu() - ref).abs().mean()), float((chained.double().cpu() - ref).abs().mean())
    assert e1[0] <= e1[1], e1
    segs, ref = [], 0
    for K in (128, 192):
        a, b = r(K, M), r(K, N)
        segs.append((a, b))
        ref = ref + a.double().cpu().t() @ b.double().cpu()
    _close(F.gemm(segs, True, True), ref, atol=float(ref.abs().max()) * 2 ** -8, rtol=2 ** -7, what="2-segment wgrad")
    with pytest.raises(_C.KernelError):
        F.gemm([(r(64, 100), r(64, 128))], True, True)       # contraction-major A: M % 8
    with pytest.raises(_C.KernelError):
        F.gemm([(r(128, 96), r(128, 96))])                   # K % 64
    with pytest.raises(_C.KernelError):
        F.gemm([(r(128, 64), r(64, 128))], True, False)      # (A contraction-major, B not): no such product
    with pytest.raises(_C.KernelError):
        F.gemm([(r(128, 64), r(128, 64))] * 4)               # more than three segments


def test_gemm_tn_strided_operands_and_rejects():
    F = _f()
    from touchnet_amd import _C
    g = torch.Generator().manual_seed(5)
    big_a = torch.randn(512, 640, generator=g).to(torch.bfloat16).to(DEV)
    big_b = torch.randn(384, 640, generator=g).to(torch.bfloat16).to(DEV)
    a, b = big_a[:, 128:640], big_b[:, :512]                      # row stride 640, K = 512, 16-byte aligned starts
    out_full = torch.zeros(512, 512, dtype=torch.bfloat16, device=DEV)
    out = out_full[:, 64:448]                                      # ldc = 512
    F.gemm_tn(a, b, out=out)
    ref = a.double().cpu() @ b.double().cpu().t()
    _close(out, ref, atol=float(ref.abs().max()) * 2 ** -8, rtol=2 ** -7, what="gemm_tn strided")
    assert float(out_full[:, :64].abs().max()) == 0 and float(out_full[:, 448:].abs().max()) == 0
    with pytest.raises(_C.KernelError):
        F.gemm_tn(big_a[:, :100], big_b[:, :100])                 # K % 64 != 0
    with pytest.raises(_C.KernelError):
        F.gemm_tn(big_a.float(), big_b.float())


def test_gemm_tn_deterministic():
    F = _f()
    g = torch.Generator().manual_seed(11)
    a = torch.randn(1024, 2048, generator=g).to(torch.bfloat16).to(DEV)
    b = torch.randn(1536, 2048, generator=g).to(torch.bfloat16).to(DEV)
    first = F.gemm_tn(a, b).clone()
    for _ in range(5):
        assert torch.equal(F.gemm_tn(a, b), first)


# ---------------------------------------------------------------------------------------- device-side packers
def _split_tokens(flat, lens):
    out, o = [], 0
    for n in lens:
        out.append([int(v) for v in flat[o:o + n]])
        o += n
    return out


@pytest.mark.parametrize("window", [3, 7, 4096])
@pytest.mark.parametrize("case", ["overflow", "exactfit", "droplast", "single"])
def test_device_packer_text_bit_exact_vs_reference_goldens(golden, case, window):
    """tn_pack_plan + tn_pack_fill == the reference's batch_text on its own golden streams, for any window size
    (open batches are carried across windows)."""
    import types
    from touchnet_amd.data.device_packer import batch_text_device
    g = golden("packing_text.npz")
    B, T, drop, nb = [int(v) for v in g[f"{case}/meta"]]
    sents = _split_tokens(g[f"{case}/tokens"], g[f"{case}/lens"])
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataloader_drop_last_batch=bool(drop))
    tok = types.SimpleNamespace(bos=1, eos=2, pad=0)
    got = list(batch_text_device(({"input_ids": s} for s in sents), cfg, tok, window=window))
    assert len(got) == nb
    for i, b in enumerate(got):
        for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens"):
            assert b[k].is_cuda and b[k].dtype == torch.int64
            np.testing.assert_array_equal(b[k].cpu().numpy(), g[f"{case}/b{i}/{k}"], err_msg=f"{case} b{i} {k}")
        assert int(b["num_sentence"]) == int(g[f"{case}/b{i}/num_sentence"])


@pytest.mark.parametrize("window", [2, 5, 1024])
@pytest.mark.parametrize("case", ["mixed", "droplast"])
def test_device_packer_asr_bit_exact_vs_reference_goldens(golden, case, window):
    import types
    from touchnet_amd.data.device_packer import batch_pairaudio_pairtext_packed_device
    g = golden("packing_asr.npz")
    B, T, drop, nb, F = [int(v) for v in g[f"{case}/meta"]]
    ids = _split_tokens(g[f"{case}/tokens"], g[f"{case}/tlens"])
    feats, o = [], 0
    for a in g[f"{case}/alens"]:
        feats.append(torch.from_numpy(g[f"{case}/feats"][o:o + a]).to(DEV))
        o += a
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                audiofeat_num_mel_bins=F, audiofeat_stack_length=1,
                                dataloader_drop_last_batch=bool(drop))
    tok = types.SimpleNamespace(bos=1, eos=2, pad=0)
    data = ({"audiofeat": f, "input_ids": i} for f, i in zip(feats, ids))
    got = list(batch_pairaudio_pairtext_packed_device(data, cfg, tok, window=window))
    assert len(got) == nb
    for i, b in enumerate(got):
        for k in ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens", "input_features",
                  "shift_labels"):
            np.testing.assert_array_equal(b[k].cpu().numpy(), g[f"{case}/b{i}/{k}"], err_msg=f"{case} b{i} {k}")
        assert int(b["num_sentence"]) == int(g[f"{case}/b{i}/num_sentence"])


@pytest.mark.parametrize("seed", range(6))
def test_device_packers_equal_host_packers_on_random_streams(seed):
    """Random B, T, drop_last, lengths incl. exact fits, empty sentences, over-long utterances (skipped) and streams
    ending on a full buffer: device packers == host packers (which the reference goldens pin), bit for bit."""
    import types
    from touchnet_amd.data.device_packer import batch_pairaudio_pairtext_packed_device, batch_text_device
    from touchnet_amd.models.llama.processing_llama import batch_text
    from touchnet_amd.models.touch_audio.processing_touch_audio import batch_pairaudio_pairtext_packed
    rng = np.random.RandomState(500 + seed)
    B, T = int(rng.randint(1, 5)), int(rng.choice([8, 17, 32, 64, 257]))
    drop = bool(rng.randint(2))
    n = int(rng.randint(0, 60))
    tok = types.SimpleNamespace(bos=1, eos=2, pad=0)
    sents = [[int(v) for v in rng.randint(3, 50, size=int(rng.randint(0, T)))] for _ in range(n)]
    cfg = types.SimpleNamespace(dataset_batchsize=B, dataset_text_seqlen=T, dataset_audio_seqlen=T,
                                audiofeat_num_mel_bins=3, audiofeat_stack_length=2, dataloader_drop_last_batch=drop)
    keys = ("input_ids", "labels", "position_ids", "attention_mask", "sentence_lens")
    want = list(batch_text(({"input_ids": s} for s in sents), cfg, tok))
    got = list(batch_text_device(({"input_ids": s} for s in sents), cfg, tok, window=int(rng.randint(1, 20))))
    assert len(got) == len(want)
    for a, b in zip(got, want):
        for k in keys:
            assert torch.equal(a[k].cpu(), b[k]), k
        assert int(a["num_sentence"]) == b["num_sentence"]
    pairs = [(torch.from_numpy(rng.randn(int(rng.randint(1, T + 3)), 6).astype(np.float32)),
              [int(v) for v in rng.randint(3, 50, size=int(rng.randint(0, 6)))]) for _ in range(n)]
    want = list(batch_pairaudio_pairtext_packed(({"audiofeat": f, "input_ids": s} for f, s in pairs), cfg, tok))
    got = list(batch_pairaudio_pairtext_packed_device(({"audiofeat": f.to(DEV), "input_ids": s} for f, s in pairs), cfg,
                                                      tok, window=int(rng.randint(1, 20))))
    assert len(got) == len(want)
    for a, b in zip(got, want):
        for k in keys + ("input_features",):
            assert torch.equal(a[k].cpu(), b[k]), k
        assert int(a["num_sentence"]) == b["num_sentence"]


def test_fused_linear_ce_static_compact_rows_equals_full_and_fails_loudly():
    """lm_head + CE on the labelled rows only, with the loader's static upper bound (no host sync): same loss, stats and
    gradients as the all-rows path; a bound that is too small turns the loss into NaN instead of dropping labels."""
    F = _f()
    g = torch.Generator().manual_seed(3)
    n, H, V = 1024, 256, 3000
    h = torch.randn(2, n // 2, H, generator=g).bfloat16().to(DEV)
    w = (torch.randn(V, H, generator=g) * 0.05).bfloat16().to(DEV)
    labels = torch.full((2, n // 2), -100)
    idx = torch.randperm(n, generator=g)[:137]
    labels.view(-1)[idx] = torch.randint(0, V, (137,), generator=g)
    labels = labels.to(DEV)
    sl = torch.randint(1, 9, (2, n // 2), generator=g).to(DEV)

    def run(compact):
        hh, ww = h.clone().requires_grad_(), w.clone().requires_grad_()
        loss, stats = F.fused_linear_cross_entropy(hh, ww, labels, sl, 7, -100, 256, compact=compact)
        loss.backward()
        return loss.detach(), stats, hh.grad, ww.grad
    full = run(False)
    for bound in (137, 256, 1024, 5000):
        got = run(bound)
        assert float(got[0]) == pytest.approx(float(full[0]), rel=1e-6), bound
        torch.testing.assert_close(got[1], full[1], rtol=1e-6, atol=1e-7)
        _close(got[2], full[2], 1e-6, 1e-2, f"dh, bound {bound}")
        _close(got[3], full[3], 4e-3, 2e-2, f"dW, bound {bound}")     # (bf16 accumulation over a different chunking: 1 ulp)
    exact = run(True)
    assert float(exact[0]) == pytest.approx(float(full[0]), rel=1e-6)
    poisoned = run(100)                                  # 137 labels do not fit 100 rows
    assert torch.isnan(poisoned[0]) and torch.isnan(poisoned[1]).all()


def test_rope_on_one_tensor_alone():
    """The context-parallel attention rotates K before Q exists (K/V travel while the query path runs): a call with a
    zero-head second operand == the joint call."""
    F = _f()
    g = torch.Generator().manual_seed(2)
    B, T, D = 2, 96, 64
    q = torch.randn(B, T, 4, D, generator=g).bfloat16().to(DEV)
    k = torch.randn(B, T, 2, D, generator=g).bfloat16().to(DEV)
    pos = torch.arange(T).repeat(B, 1).to(DEV)
    cos, sin = F.rope_tables(pos, F.rope_inv_freq(D, 10000.0, None, device=DEV), torch.bfloat16)
    qj, kj = F.apply_rope(q, k, cos, sin)
    k1, e = F.apply_rope(k, k.new_empty(B, T, 0, D), cos, sin)
    q1, _ = F.apply_rope(q, q.new_empty(B, T, 0, D), cos, sin)
    assert torch.equal(k1, kj) and torch.equal(q1, qj) and e.numel() == 0


@pytest.mark.parametrize("n,T,C,O,stride", [(3, 160, 16, 128, 1), (3, 160, 128, 128, 2), (2, 3000, 128, 1280, 1),
                                            (2, 3000, 1280, 1280, 2), (1, 77, 64, 192, 2)])
def test_conv1d_k3_as_gemm_over_strided_im2col_views(n, T, C, O, stride):
    """functional.conv1d_k3 (the audio tower's conv stem on the hand-written GEMM: overlapping-row im2col views, zero-separated
    clip slots, overlap-add input gradient) against torch's conv1d in fp32 on the same bf16 values: output, d(input),
    d(weight), d(bias)."""
    F = _f()
    g = torch.Generator().manual_seed(n + T + C + O)
    x = (torch.randn(n, T, C, generator=g) * 0.5).bfloat16()
    w = (torch.randn(O, C, 3, generator=g) * (3 * C) ** -0.5).bfloat16()
    b = (torch.randn(O, generator=g) * 0.1).bfloat16()
    xr, wr, br = (t.float().requires_grad_() for t in (x, w, b))
    ref = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, br, stride=stride, padding=1).transpose(1, 2)
    dyr = torch.randn(ref.shape, generator=g) * 0.3
    ref.backward(dyr)
    xd, wd, bd = (t.to(DEV).requires_grad_() for t in (x, w, b))
    full, t_out = F.conv1d_k3(xd, wd, bd, stride)
    assert t_out == ref.shape[1] and full.shape[1] >= t_out
    y = full[:, :t_out]
    y.backward(dyr.bfloat16().to(DEV))
    _close(y, ref.detach(), atol=float(ref.abs().max()) * 2 ** -7, rtol=2 ** -6, what="conv1d_k3 forward")
    for name, got, want in (("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad), ("db", bd.grad, br.grad)):
        _close(got, want, atol=float(want.abs().max()) * 2 ** -6, rtol=2 ** -5, what=f"conv1d_k3 {name}")
    xd2 = x.to(DEV).requires_grad_()
    F.conv1d_k3(xd2, wd, bd, stride, need_dx=False)[0][:, :t_out].sum().backward()
    assert xd2.grad is None
    from touchnet_amd import _C
    with pytest.raises(_C.KernelError):
        F.conv1d_k3(xd, wd[:72], bd[:72], stride)                          # the input gradient contracts over 72 channels
