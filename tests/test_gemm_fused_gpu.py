"""Round-5 GEMM fusions on the MI355X (csrc/gemm.hip): SwiGLU inside the gate/up and down-dgrad epilogues, grouped weight
gradients.  The fused epilogues are held BIT FOR BIT to the launches they replace (same products, same rounding points,
same SwiGLU arithmetic), the grouped launch to its single products (whole tiles bitwise; the split-K remainder to fp64)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _f():
    import touchnet_amd.functional as F
    return F


def _r(g, *shape, scale=1.0):
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(torch.bfloat16).to(DEV)


@pytest.mark.parametrize("M,I,K", [
    (256, 128, 64),          # one tile (128 gate + 128 up columns), one stage
    (520, 264, 192),         # ragged rows; I % 128 != 0: the last tile's up / gate panels end inside a 32-row DMA run
    (1000, 776, 1088),       # 17 stages
    (4360, 1096, 128),       # 18 x 9 = 162 tiles ... and
    (8200, 1096, 128),       # 33 x 9 = 297 tiles > 256 CUs: persistent workgroups, park in the freed slots
    (2048, 11008, 4096),     # the 7B MLP's shape (86 column tiles)
])
def test_swiglu_forward_epilogue_is_bit_identical_to_two_products_and_the_swiglu_kernel(M, I, K):
    F = _f()
    g = torch.Generator().manual_seed(M + 3 * I + 7 * K)
    x, wg, wu = _r(g, M, K), _r(g, I, K, scale=K ** -0.5 * 2), _r(g, I, K, scale=K ** -0.5 * 2)
    gate, up, act = F.gemm_swiglu_fwd(x, wg, wu)
    g_ref, u_ref = F.gemm([(x, wg)]), F.gemm([(x, wu)])
    a_ref = F.swiglu(g_ref, u_ref)
    assert torch.equal(gate, g_ref), "gate"
    assert torch.equal(up, u_ref), "up"
    assert torch.equal(act, a_ref), "act"
    # and the products themselves against fp64 (the fused tile map must not have permuted columns)
    ref = x.double() @ wg.double().t()
    assert float((gate.double() - ref).abs().max()) <= float(ref.abs().max()) * 2 ** -7


@pytest.mark.parametrize("M,I,H", [
    (256, 256, 64),
    (520, 264, 192),
    (1000, 776, 1088),
    (4360, 4104, 128),       # 306 tiles: persistent
    (2048, 11008, 4096),     # the 7B MLP's d(act) product
])
def test_swiglu_backward_epilogue_is_bit_identical_to_the_product_and_the_swiglu_kernel(M, I, H):
    F = _f()
    from touchnet_amd import library as L
    g = torch.Generator().manual_seed(M + 5 * I + 11 * H)
    dy, wd = _r(g, M, H), _r(g, H, I, scale=H ** -0.5 * 2)
    gate, up = _r(g, M, I, scale=3.0), _r(g, M, I)
    dgate, dup = F.gemm_swiglu_bwd(dy, wd, gate, up)
    dact = F.gemm([(dy, wd)], b_kmaj=True)
    dg_ref, du_ref = L.swiglu_bwd(dact, gate, up)
    assert torch.equal(dgate, dg_ref), "d(gate)"
    assert torch.equal(dup, du_ref), "d(up)"


def test_swiglu_epilogues_reject_what_the_kernel_cannot_take():
    F = _f()
    from touchnet_amd import _C
    z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_C.KernelError):
        F.gemm_swiglu_fwd(z(64, 100), z(128, 100), z(128, 100))            # K % 64
    with pytest.raises(_C.KernelError):
        F.gemm_swiglu_fwd(z(64, 128), z(132, 128), z(132, 128))            # I % 8
    with pytest.raises(_C.KernelError):
        F.gemm_swiglu_bwd(z(64, 128), z(128, 256), z(64, 256), z(64, 128))  # gate / up shapes differ


@pytest.mark.parametrize("shapes,K", [
    ([(256, 256), (512, 256)], 192),                          # 1 + 2 tiles
    ([(520, 264), (264, 520), (1000, 776)], 1000),            # ragged, depth not a multiple of 64 (zero-filled tail stage)
    ([(2816, 1024), (2816, 1024), (1024, 2816)], 2048),       # 3 x 44 tiles = 132: no whole round, no split
    ([(11008, 4096), (11008, 4096), (4096, 11008)], 2048),    # the MLP: 2064 tiles = 8 rounds + 16 tiles in 4 parts
])
def test_grouped_weight_gradients_equal_the_single_products(shapes, K):
    """One launch over the concatenated tile lists: tiles of the whole rounds are computed exactly as in a single-product
    launch (bitwise); the split-K remainder differs in summation order only (fp64 bound)."""
    F = _f()
    g = torch.Generator().manual_seed(sum(m + n for m, n in shapes) + K)
    pairs = [(_r(g, K, m), _r(g, K, n)) for m, n in shapes]
    outs = F.gemm_grouped_wgrad(pairs)
    tiles = sum(-(-m // 256) * -(-n // 256) for m, n in shapes)
    split_tail = tiles > 256 and 0 < tiles % 256 <= 128 and K >= 1024
    n_diff = 0
    for (a, b), o in zip(pairs, outs):
        ref = a.double().t() @ b.double()
        assert tuple(o.shape) == tuple(ref.shape)
        err = float((o.double() - ref).abs().max())
        assert err <= float(ref.abs().max()) * 2 ** -7, err
        if F.split_k(o.shape[0], o.shape[1], K, True, True) == 1:      # (a lone product of few tiles is split-K itself)
            n_diff += int((o != F.gemm([(a, b)], True, True)).sum())
    if not split_tail:
        assert n_diff == 0
    else:
        assert n_diff <= (tiles % 256) * 65536            # only inside the remainder's tiles
    again = F.gemm_grouped_wgrad(pairs)
    assert all(torch.equal(x, y) for x, y in zip(outs, again))          # deterministic (no atomics)


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("shapes,K", [([(2560, 2560)] * 3, 2048),                # 300 tiles: 256 whole + 44 in 4 parts
                                      ([(2816, 2560), (2560, 2816), (2816, 2560)], 1536),   # 330: remainder 74 in 3 parts
                                      ([(4096, 4096), (2304, 256), (256, 2304)], 1024)])    # remainder spans TWO products
def test_grouped_weight_gradients_with_a_split_remainder(shapes, K, f32):
    """the remainder of the tile list (tiles mod 256) runs split-K per product it touches: bf16 and fp32 outputs vs fp64"""
    F = _f()
    g = torch.Generator().manual_seed(17 + K)
    pairs = [(_r(g, K, m), _r(g, K, n)) for m, n in shapes]
    outs = [torch.full((m, n), float("nan"), dtype=torch.float32 if f32 else torch.bfloat16, device=DEV) for m, n in shapes]
    F.gemm_grouped_wgrad(pairs, outs=outs)
    for i, ((a, b), o) in enumerate(zip(pairs, outs)):
        ref = a.double().t() @ b.double()
        err = (o.double() - ref).abs()
        assert not torch.isnan(o).any(), (i, int(torch.isnan(o).sum()))
        assert float(err.max()) <= float(ref.abs().max()) * (2e-5 if f32 else 2 ** -7), (i, float(err.max()))


def test_grouped_weight_gradients_into_fp32_buffers_with_accumulation():
    """the data-parallel engine's form: fp32 outputs (views of a flat staging buffer), C += dY^T x"""
    F = _f()
    g = torch.Generator().manual_seed(5)
    K = 1536
    shapes = [(2816, 1024), (2816, 1024), (1024, 2816)]
    pairs = [(_r(g, K, m), _r(g, K, n)) for m, n in shapes]
    flat = torch.randn(sum(m * n for m, n in shapes), device=DEV)
    views, o = [], 0
    for m, n in shapes:
        views.append(flat[o:o + m * n].view(m, n))
        o += m * n
    before = [v.clone() for v in views]
    F.gemm_grouped_wgrad(pairs, outs=views, accumulate=True)
    for (a, b), v, v0 in zip(pairs, views, before):
        ref = v0.double() + a.double().t() @ b.double()
        assert float((v.double() - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
    F.gemm_grouped_wgrad(pairs, outs=views)                       # overwrite: the unrounded fp32 accumulators
    for (a, b), v in zip(pairs, views):
        ref = a.double().t() @ b.double()
        assert float((v.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("M,H,I", [(8192, 1024, 2816), (4096, 2048, 5632)])
def test_mlp_node_with_fused_epilogues_and_grouped_wgrads_equals_the_unfused_node(M, H, I, monkeypatch):
    """`swiglu_mlp` end to end (autograd): fused epilogues + grouped weight gradients against the round-4 composition of
    seven GEMM launches and two SwiGLU kernels — output, d(x) bit-identical; weight gradients bit-identical when no
    remainder is split, else to fp32 summation order."""
    F = _f()
    g = torch.Generator().manual_seed(M + H)
    x = _r(g, 2, M // 2, H)
    ws = [_r(g, n, k, scale=k ** -0.5 * 2) for n, k in ((I, H), (I, H), (H, I))]
    dy = _r(g, 2, M // 2, H)

    def run():
        xx = x.clone().requires_grad_()
        ww = [w.clone().requires_grad_() for w in ws]
        y = F.swiglu_mlp(xx, *ww)
        y.backward(dy)
        return [y.detach(), xx.grad] + [w.grad for w in ww]

    calls = {"fwd": 0, "bwd": 0, "grp": 0}
    for key, name in (("fwd", "gemm_swiglu_fwd"), ("bwd", "gemm_swiglu_bwd"), ("grp", "gemm_grouped_wgrad")):
        orig = getattr(F, name)
        monkeypatch.setattr(F, name, (lambda o, k: lambda *a, **kw: (calls.__setitem__(k, calls[k] + 1), o(*a, **kw))[1])(orig, key))
    monkeypatch.setattr(F, "MLP_EPILOGUE", True)
    monkeypatch.setattr(F, "GROUPED_WGRAD", True)
    got = run()
    assert calls == {"fwd": 1, "bwd": 1, "grp": 1}, calls           # the fused path is the one that ran
    monkeypatch.setattr(F, "MLP_EPILOGUE", False)
    monkeypatch.setattr(F, "GROUPED_WGRAD", False)
    want = run()
    for name, a, b in zip(("y", "dx", "dWgate", "dWup", "dWdown"), got, want):
        if name in ("y", "dx"):
            assert torch.equal(a, b), name
        else:
            torch.testing.assert_close(a.float(), b.float(), rtol=2 ** -6, atol=2 ** -7 * float(b.abs().max()),
                                       msg=lambda m: f"{name}: {m}")


def test_bitwise_repeatability_of_split_k_and_grouped_products():
    """a race shows up as ONE different launch in hundreds — 200 launches each of the split-K weight
    gradient (tower shape), the grouped MLP weight gradients and both SwiGLU epilogues must be bitwise equal to the first."""
    F = _f()
    g = torch.Generator().manual_seed(99)
    a, b = _r(g, 3000, 1280), _r(g, 3000, 1280)
    first = F.gemm([(a, b)], True, True)
    pairs = [(_r(g, 1024, 2816), _r(g, 1024, 1024)), (_r(g, 1024, 2816), _r(g, 1024, 1024)),
             (_r(g, 1024, 1024), _r(g, 1024, 2816))]
    first_g = F.gemm_grouped_wgrad(pairs)
    x, wg, wu = _r(g, 1024, 512), _r(g, 1376, 512, scale=0.1), _r(g, 1376, 512, scale=0.1)
    first_f = F.gemm_swiglu_fwd(x, wg, wu)
    dy, wd = _r(g, 1024, 512), _r(g, 512, 1376, scale=0.1)
    first_b = F.gemm_swiglu_bwd(dy, wd, first_f[0], first_f[1])
    for _ in range(200):
        assert torch.equal(F.gemm([(a, b)], True, True), first)
        assert all(torch.equal(u, v) for u, v in zip(F.gemm_grouped_wgrad(pairs), first_g))
        assert all(torch.equal(u, v) for u, v in zip(F.gemm_swiglu_fwd(x, wg, wu), first_f))
        assert all(torch.equal(u, v) for u, v in zip(F.gemm_swiglu_bwd(dy, wd, first_f[0], first_f[1]), first_b))


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("M,N,K", [
    (256, 256, 64),           # one tile, one stage
    (520, 264, 1000),         # ragged rows / columns / depth (zero-filled tail stage contributes nothing)
    (4360, 520, 192),         # 18 x 3 tiles: only the tiles of output column 0 carry the sums
    (4096, 4096, 2048),       # the decoder's q/k/v shape: 256 tiles, one round
    (6200, 4104, 128),        # 425 tiles: persistent workgroups meet several column-0 tiles
    (1280, 1280, 30000),      # audio tower q / v / out_proj: 25 tiles, split-K in 10 parts (partials through the workspace)
    (5120, 1280, 30000),      # fc1: 100 tiles, 2 parts
])
def test_weight_gradient_launch_returns_the_bias_gradient(M, N, K, f32):
    """EPI_BIASG: dW = dY^T x and db = dY.sum(0) from ONE launch; dW bit-identical to the launch without the bias output,
    db against an fp64 column sum of the same bf16 values (fp32 accumulation, one bf16 rounding)."""
    F = _f()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    dy, x = _r(g, K, M), _r(g, K, N)
    dy[:, :8] += 0.5                                              # (a non-zero mean: the sum is not just noise)
    kw = dict(out=torch.empty(M, N, dtype=torch.float32, device=DEV)) if f32 else {}
    db = torch.full((M,), float("nan"), dtype=torch.bfloat16, device=DEV)
    dw = F.gemm([(dy, x)], True, True, bias_grad=db, **kw).clone()
    kw = dict(out=torch.empty(M, N, dtype=torch.float32, device=DEV)) if f32 else {}
    ref_w = F.gemm([(dy, x)], True, True, **kw)
    assert torch.equal(dw, ref_w)
    ref = dy.double().sum(0)
    tol = 2.0 ** -8 * ref.abs().clamp_min(1.0) + 1e-3 * K ** 0.5
    assert not torch.isnan(db).any()
    assert ((db.double() - ref).abs() <= tol).all(), float((db.double() - ref).abs().max())
    db2 = torch.empty_like(db)
    F.gemm([(dy, x)], True, True, bias_grad=db2, **kw)
    assert torch.equal(db, db2)                                   # deterministic


def test_linear_group_bias_gradients_come_from_the_weight_gradient_launch(monkeypatch):
    """autograd through `linear_group` (q/k/v with biases at the decoder's width): no column-sum launch is left, the bias
    gradients equal the separate pass to bf16 rounding of differently ordered fp32 sums"""
    F = _f()
    g = torch.Generator().manual_seed(3)
    M, H = 4096, 4096
    x = _r(g, M, H, scale=0.5)
    layers = [(_r(g, n, H, scale=H ** -0.5), _r(g, n, scale=0.1)) for n in (4096, 512, 512)]
    dys = [_r(g, M, n) for n in (4096, 512, 512)]

    def run():
        xx = x.clone().requires_grad_()
        lw = [(w.clone().requires_grad_(), b.clone().requires_grad_()) for w, b in layers]
        outs = F.linear_group(xx, lw)
        torch.autograd.backward(outs, dys)
        return [xx.grad] + [w.grad for w, _ in lw] + [b.grad for _, b in lw]

    calls = []
    orig = F.column_sum
    monkeypatch.setattr(F, "column_sum", lambda t: (calls.append(1), orig(t))[1])
    monkeypatch.setattr(F, "BIAS_IN_WGRAD", True)
    got = run()
    n_fused = len(calls)
    monkeypatch.setattr(F, "BIAS_IN_WGRAD", False)
    want = run()
    assert n_fused == 0 and len(calls) == 3
    for a, b in zip(got[:4], want[:4]):
        assert torch.equal(a, b)                                  # dx and the weight gradients are untouched
    for a, b, dy in zip(got[4:], want[4:], dys):
        ref = dy.double().sum(0)
        assert float((a.double() - ref).abs().max()) <= float((b.double() - ref).abs().max()) + 2.0 ** -7 * float(ref.abs().max())


def test_bitwise_repeatability_of_the_cross_entropy_kernels():
    """VERDICT r4 #13, CE part: 200 launches of the packed CE forward + backward at the headline's vocabulary (156032) and of
    the fused lm_head + CE (chunked, labelled rows) return the SAME bits as the first — statistics, loss, d(logits),
    d(hidden), d(weight)."""
    F = _f()
    g = torch.Generator().manual_seed(7)
    V, n, H = 156032, 96, 512
    logits = (torch.randn(1, n, V, generator=g) * 2).to(torch.bfloat16).to(DEV)
    labels = torch.randint(0, V, (1, n), generator=g)
    labels[0, ::5] = -100
    labels = labels.to(DEV)
    sl = torch.randint(1, 9, (1, n), generator=g).to(DEV)
    hidden = (torch.randn(1, n, H, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w = (torch.randn(4096, H, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    lab2 = (labels % 4096).where(labels >= 0, labels)

    def once():
        lg = logits.clone().requires_grad_()
        loss, stats = F.packed_cross_entropy(lg, labels, sl, 12)
        loss.backward()
        hh, ww = hidden.clone().requires_grad_(), w.clone().requires_grad_()
        l2, st2 = F.fused_linear_cross_entropy(hh, ww, lab2, sl, 12, chunk_tokens=32, compact=128)
        l2.backward()
        return [loss.detach(), stats, lg.grad, l2.detach(), st2, hh.grad, ww.grad]

    first = once()
    assert all(bool(torch.isfinite(t.float()).all()) for t in first)
    for _ in range(200):
        for a, b in zip(once(), first):
            assert torch.equal(a, b)


@pytest.mark.parametrize("M,heads,D,K,bias", [
    (256, 2, 128, 64, True),          # one tile = two heads
    (520, 6, 128, 192, True),         # ragged rows, three column tiles
    (4360, 8, 128, 128, False),       # 18 x 4 tiles, no bias
    (8200, 8, 128, 128, True),        # 264 tiles: persistent workgroups
    (4360, 32, 128, 4096, True),      # the 7B q / k projection (272 tiles: the unfused reference is not split-K)
    (1000, 12, 64, 256, True),        # head_dim 64 (Llama-3.2-1B, config B): 4 heads per tile, natural pair layout
    (300, 5, 64, 128, False),         # N = 320: a ragged last column tile of whole heads
])
def test_rope_epilogue_is_bit_identical_to_projection_plus_rope_kernel(M, heads, D, K, bias):
    """tn_gemm_bf16_rope: rope(x W^T + b) from one launch == tn_gemm_bf16 followed by tn_rope_apply, bit for bit (the
    product is rounded to bf16 where the separate projection rounds it, the rotation is the row kernel's `rope_rotate`)."""
    F = _f()
    g = torch.Generator().manual_seed(M + heads + D + K)
    x, w = _r(g, M, K), _r(g, heads * D, K, scale=K ** -0.5 * 2)
    b = _r(g, heads * D) if bias else None
    pos = torch.randint(0, 4000, (1, M), generator=g).to(DEV)
    cos, sin = F.rope_tables(pos, F.rope_inv_freq(D, 1e6, device=DEV), torch.bfloat16)
    got = F.gemm_rope(x, w, b, cos, sin, D)
    y = F.gemm([(x, w)], bias=b).view(1, M, heads, D)
    ref = F.apply_rope(y, y.new_empty(1, M, 0, D), cos, sin)[0].view(M, heads * D)
    assert torch.equal(got, ref)
    # against fp64 as well: a wrong pair / column permutation cannot hide behind the self-comparison
    yd = (x.double() @ w.double().t() + (b.double() if bias else 0)).view(M, heads, D)
    c = torch.cat([cos, cos], -1).double()[:, None, :]
    sgn = torch.cat([-yd[..., D // 2:], yd[..., :D // 2]], -1)
    rd = (yd * c + sgn * torch.cat([sin, sin], -1).double()[:, None, :]).view(M, heads * D)
    assert float((got.double() - rd).abs().max()) <= float(rd.abs().max()) * 2 ** -6


def test_linear_group_with_rope_outputs_trains_like_projection_then_rope(monkeypatch):
    """autograd: q, k rotated in the projections' epilogues (v plain), gradients rotated back in front of the products —
    outputs and every gradient bit-identical to linear_group followed by apply_rope"""
    F = _f()
    g = torch.Generator().manual_seed(11)
    M, H, D = 8192, 1024, 128
    x = _r(g, 1, M, H, scale=0.5)
    layers = [(_r(g, n, H, scale=H ** -0.5), _r(g, n, scale=0.1)) for n in (2048, 1280, 512)]    # 256 / 160 / 64 tiles
    dys = [_r(g, 1, M, n) for n in (2048, 1280, 512)]
    pos = torch.arange(M)[None].to(DEV)
    cos, sin = F.rope_tables(pos, F.rope_inv_freq(D, 1e6, device=DEV), torch.bfloat16)

    def run(fused):
        monkeypatch.setattr(F, "ROPE_EPILOGUE", fused)
        xx = x.clone().requires_grad_()
        lw = [(w.clone().requires_grad_(), b.clone().requires_grad_()) for w, b in layers]
        if fused is None:                                         # the round-4 composition
            q, k, v = F.linear_group(xx, lw)
            q, k = F.apply_rope(q.view(1, M, -1, D), k.view(1, M, -1, D), cos, sin)
            outs = [q.reshape(1, M, -1), k.reshape(1, M, -1), v]
        else:
            outs = F.linear_group(xx, lw, rope=(cos, sin, D, (0, 1)))
        torch.autograd.backward(outs, dys)
        return [o.detach() for o in outs] + [xx.grad] + [w.grad for w, _ in lw] + [b.grad for _, b in lw]

    calls = []
    orig = F.gemm_rope
    monkeypatch.setattr(F, "gemm_rope", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    want = run(None)
    got = run(True)
    assert len(calls) == 2
    behind = run(False)                                           # epilogue switched off: rotated behind the product
    assert len(calls) == 2
    for i, (a, b, c) in enumerate(zip(got, want, behind)):
        assert torch.equal(a, b) and torch.equal(c, b), i


@pytest.mark.parametrize("heads,kv_heads", [(8, 8), (8, 2)])
def test_attention_block_with_the_rotary_gradient_in_the_attention_backward_trains_bit_identically(heads, kv_heads, monkeypatch):
    """models/llama Attention (q / k / v projections with the rotary embedding in their epilogues -> packed attention ->
    o_proj): with functional.ROPE_GRAD_IN_ATTENTION the attention backward returns dq / dk rotated back (tn_attn_bwd_rope) and
    the projection node skips its pass over them — output and every gradient equal the separate-pass run bit for bit
    (multi-head: stacked gradient buffer; grouped-query: flat buffer)."""
    F = _f()
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.models.llama.modeling_llama import Attention
    cfg = DecoderConfig.from_dict(dict(model_type="qwen2", hidden_size=1024, intermediate_size=2048, num_attention_heads=heads,
                                       num_key_value_heads=kv_heads, head_dim=128, num_hidden_layers=1, vocab_size=64,
                                       attention_bias=True))
    torch.manual_seed(5)
    att = Attention(cfg).to(DEV).to(torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    B, T = 2, 1024
    x = _r(g, B, T, 1024, scale=0.5)
    dy = _r(g, B, T, 1024)
    doc = torch.ones(B, T, dtype=torch.int32)
    doc[0, 500:] = 2
    doc[1, 900:] = 0
    mask = F.build_packed_mask(doc.to(DEV))
    pos = torch.arange(T)[None].expand(B, T).contiguous().to(DEV)
    cos, sin = F.rope_tables(pos, F.rope_inv_freq(128, 1e6, device=DEV), torch.bfloat16)
    calls = []
    orig = F._AttentionRopeGrad.apply
    monkeypatch.setattr(F._AttentionRopeGrad, "apply", lambda *a: (calls.append(1), orig(*a))[1])

    def run(on):
        monkeypatch.setattr(F, "ROPE_GRAD_IN_ATTENTION", on)
        xx = x.clone().requires_grad_()
        for p in att.parameters():
            p.grad = None
        y = att(xx, cos, sin, mask)
        y.backward(dy)
        return [y.detach(), xx.grad] + [p.grad.clone() for p in att.parameters()]

    want = run(False)
    assert not calls
    got = run(True)
    assert len(calls) == 1
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))


@pytest.mark.parametrize("M,N,K,bias,kmaj", [
    (256, 256, 64, False, False), (520, 264, 192, True, False), (8200, 1096, 128, False, False),
    (15872, 4096, 4096, False, False),      # o_proj at the headline's rows
    (4096, 4096, 11008, False, False),      # down_proj (256 tiles: not a split-K shape, whose fp32 partial sums round differently)
    (30000, 1280, 5120, True, False),       # the tower's fc2 (bias) at the headline's frames
    (1000, 776, 1088, False, True),         # contraction-major B (32x32x16 kernel's epilogue... the 16x16 one takes it too)
])
def test_residual_addend_epilogue_is_bit_identical_to_the_product_plus_the_norm_kernels_addition(M, N, K, bias, kmaj):
    """Round 6: `hidden_states = residual + hidden_states` in the epilogue of the producing GEMM (tn_gemm_bf16_addend).  The
    product is rounded to bf16 and added to the addend in fp32 — exactly what GEMM + the fused norm kernel's residual add
    did: bits must agree with `rms_norm(delta, residual)`'s second output and with bf16(a + b)."""
    F = _f()
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = _r(g, M, K)
    b = _r(g, K, N, scale=K ** -0.5 * 2) if kmaj else _r(g, N, K, scale=K ** -0.5 * 2)
    r = _r(g, M, N, scale=2.0)
    bv = _r(g, N) if bias else None
    y = F.gemm([(a, b)], b_kmaj=kmaj, bias=bv, addend=r)
    ref = F.gemm([(a, b)], b_kmaj=kmaj, bias=bv)
    assert torch.equal(y, (ref.float() + r.float()).to(torch.bfloat16))
    if N % 8 == 0 and not kmaj:
        _, h = F.rms_norm(ref, torch.ones(N, dtype=torch.bfloat16, device=DEV), 1e-6, residual=r)
        assert torch.equal(y, h)
    # refused: aliasing the output, a second segment, accumulate
    with pytest.raises(Exception):
        F.gemm([(a, b)], b_kmaj=kmaj, out=r, addend=r)


@pytest.mark.parametrize("M,N,K,bias", [(256, 256, 64, True), (520, 264, 192, False), (8200, 1096, 128, True),
                                        (30000, 5120, 1280, True)])        # the tower's fc1 at the headline's frames
def test_gelu_forward_epilogue_is_bit_identical_to_the_product_and_the_gelu_kernel(M, N, K, bias):
    F = _f()
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    x, w = _r(g, M, K), _r(g, N, K, scale=K ** -0.5 * 4)
    b = _r(g, N) if bias else None
    pre, act = F.gemm_gelu_fwd(x, w, b)
    ref = F.gemm([(x, w)], bias=b)
    assert torch.equal(pre, ref), "pre"
    assert torch.equal(act, F.gelu(ref)), "act"
    # the shared GELU itself against torch's exact-erf GELU in fp64 (A&S 7.1.26: <= 1.5e-7 on erf)
    exact = torch.nn.functional.gelu(ref.double())
    assert float((act.double() - exact).abs().max()) <= float(exact.abs().max()) * 2 ** -8


@pytest.mark.parametrize("M,I,H", [(256, 256, 64), (520, 264, 192), (4360, 4104, 128), (30000, 5120, 1280)])
def test_gelu_backward_epilogue_is_bit_identical_to_the_product_and_the_gelu_backward_kernel(M, I, H):
    F = _f()
    from touchnet_amd import library as L
    g = torch.Generator().manual_seed(M + 5 * I + 11 * H)
    dy, w2 = _r(g, M, H), _r(g, H, I, scale=H ** -0.5 * 2)
    pre = _r(g, M, I, scale=3.0)
    dpre = F.gemm_gelu_bwd(dy, w2, pre)
    dact = F.gemm([(dy, w2)], b_kmaj=True)
    assert torch.equal(dpre, L.gelu_bwd(dact, pre))
    # gelu' against autograd of torch's exact GELU
    xg = pre.double().requires_grad_()
    torch.nn.functional.gelu(xg).backward(dact.double())
    assert float((dpre.double() - xg.grad).abs().max()) <= float(xg.grad.abs().max()) * 2 ** -7


def test_tower_layer_fused_mlp_equals_the_unfused_layer(monkeypatch):
    """functional._GeluMLP (GELU in fc1's epilogue, its backward in fc2's input-gradient epilogue) against the composition
    of the individual ops: same outputs, same gradients, bit for bit."""
    F = _f()
    g = torch.Generator().manual_seed(5)
    M, d, ffn = 6000, 1280, 5120
    x = _r(g, 1, M, d).requires_grad_()
    w1, b1 = _r(g, ffn, d, scale=0.05).requires_grad_(), _r(g, ffn, scale=0.1).requires_grad_()
    w2, b2 = _r(g, d, ffn, scale=0.03).requires_grad_(), _r(g, d, scale=0.1).requires_grad_()
    dy = _r(g, 1, M, d)

    def run():
        for t in (x, w1, b1, w2, b2):
            t.grad = None
        y = F.gelu_mlp(x, w1, b1, w2, b2)
        y.backward(dy)
        return [y.detach().clone()] + [t.grad.clone() for t in (x, w1, b1, w2, b2)]
    a = run()
    monkeypatch.setattr(F, "GELU_EPILOGUE", False)
    b = run()
    for name, u, v in zip(("y", "dx", "dw1", "db1", "dw2", "db2"), a, b):
        assert torch.equal(u, v), name
