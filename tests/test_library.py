"""`torch.ops.mi355_touch.*` (touchnet_amd/library.py): schemas, meta kernels, autograd registration — SURVEY §8b "C-ABI
realisation".  CPU part: everything that must work WITHOUT the device (registration, fake tensors, meta init); the GPU
part runs torch.library.opcheck on the real kernels."""
import pytest
import torch

import touchnet_amd.library as L


def test_every_op_is_registered_with_schema_and_fake_kernel():
    for name in L.OPS:
        op = getattr(torch.ops.mi355_touch, name)
        schema = op.default._schema
        assert str(schema).startswith(f"mi355_touch::{name}("), schema
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"mi355_touch::{name}", "Meta"), name
    assert "Tensor? residual" in str(torch.ops.mi355_touch.rmsnorm_fwd.default._schema)
    assert "SymInt[] segs" in str(torch.ops.mi355_touch.attn_fwd_seg.default._schema)
    assert "Tensor(a0!) logits" in str(torch.ops.mi355_touch.ce_bwd_.default._schema)      # declared mutation


def _tiny_model(device):
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    cfg = DecoderConfig(vocab_size=128, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=1, head_dim=64, model_type="qwen2")
    with torch.device(device):
        model = PackedCausalLM(cfg)
    for t in list(model.parameters()) + list(model.buffers()):
        if t.is_floating_point():
            t.data = t.data.to(torch.bfloat16)
    return model


def test_whole_model_forward_on_fake_cuda_tensors_without_a_gpu():
    """FakeTensorMode over 'cuda' tensors: the forward of the product model runs through the dispatcher with the
    PRODUCT op set (no oracle), shapes only — what torch.compile / export of the surrounding module needs."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        model = _tiny_model("cuda")
        B, T = 2, 128
        ids = torch.zeros(B, T, dtype=torch.int64, device="cuda")
        doc = torch.ones(B, T, dtype=torch.int64, device="cuda")
        out = model(input_ids=ids, position_ids=ids, attention_mask=doc)
        assert out.logits.shape == (B, T, 128) and out.logits.device.type == "cuda"
        assert out.logits.dtype == torch.bfloat16


def test_whole_step_on_the_meta_device():
    """Meta tensors (the device the reference trainer builds the model on, train.py:179-182): forward + packed loss +
    BACKWARD through every registered autograd formula, fused-CE branch included."""
    from touchnet_amd.loss.cross_entropy import cross_entropy_loss
    model = _tiny_model("meta")
    B, T = 2, 128
    ids = torch.zeros(B, T, dtype=torch.int64, device="meta")
    doc = torch.ones(B, T, dtype=torch.int64, device="meta")
    ns = torch.ones(1, device="meta")
    out = model(input_ids=ids, position_ids=ids, attention_mask=doc)
    loss, _ = cross_entropy_loss(out.logits, ids, doc, ns)
    loss.backward()
    for n, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n
    model.zero_grad()
    out = model(input_ids=ids, position_ids=ids, attention_mask=doc, labels=ids, sentence_lens=doc, num_sentence=ns,
                ce_chunk_tokens=64)
    assert out.logits is None and out.loss.shape == ()
    out.loss.backward()
    assert model.lm_head.weight.grad.shape == model.lm_head.weight.shape


def test_ops_are_visible_to_dispatch_modes():
    """A TorchDispatchMode sees the custom ops by name (that is what op-level activation-checkpoint policies match on,
    touchnet/models/helper_func.py:39-96)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from torch.utils._python_dispatch import TorchDispatchMode

    import touchnet_amd.functional as F
    seen = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            seen.append(str(func))
            return func(*args, **(kwargs or {}))

    with FakeTensorMode(), Spy():
        x = torch.empty(4, 16, 128, device="cuda", dtype=torch.bfloat16)
        w = torch.empty(128, device="cuda", dtype=torch.bfloat16)
        y, h = F.rms_norm(x, w, 1e-5, residual=x)
        F.swiglu(y, h)
    assert any("mi355_touch.rmsnorm_fwd" in s for s in seen) and any("mi355_touch.swiglu_fwd" in s for s in seen)


@pytest.mark.gpu
def test_opcheck_on_device():
    """torch.library.opcheck: schema correctness, fake-vs-real agreement (shapes / strides / dtypes), autograd
    registration and AOT dispatch — on the real kernels."""
    from torch.library import opcheck
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s, dt=bf, grad=False: torch.randn(*s, device=dev, generator=g).to(dt).requires_grad_(grad)
    x, w = r(6, 32, 256, grad=True), r(256, grad=True)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    opcheck(torch.ops.mi355_touch.rmsnorm_fwd.default, (x, None, w, 1e-5), test_utils=tests)
    opcheck(torch.ops.mi355_touch.rmsnorm_fwd.default, (x, r(6, 32, 256, grad=True), w, 1e-5), test_utils=tests)
    opcheck(torch.ops.mi355_touch.layernorm_fwd.default, (x, None, w, r(256, grad=True), 1e-5), test_utils=tests)
    opcheck(torch.ops.mi355_touch.swiglu_fwd.default, (r(64, 512, grad=True), r(64, 512, grad=True)), test_utils=tests)
    opcheck(torch.ops.mi355_touch.gelu_fwd.default, (r(64, 512, grad=True),), test_utils=tests)
    B, T, Nh, Nkv, D = 2, 256, 4, 2, 64
    q, k, v = r(B, T, Nh, D, grad=True), r(B, T, Nkv, D, grad=True), r(B, T, Nkv, D, grad=True)
    cos, sin = r(B * T, D // 2), r(B * T, D // 2)
    opcheck(torch.ops.mi355_touch.rope_apply.default, (q, k, cos, sin, False), test_utils=tests)
    doc = torch.ones(B, T, dtype=torch.int32, device=dev)
    doc[1, 200:] = 0
    meta = torch.ops.mi355_touch.attn_build_meta(doc)
    opcheck(torch.ops.mi355_touch.attn_build_meta.default, (doc,), test_utils=("test_schema", "test_faketensor"))
    opcheck(torch.ops.mi355_touch.attn_fwd.default, (q, k, v, doc, meta, 0.125), test_utils=tests)
    n, V = 64, 1000
    logits = r(n, V, grad=True)
    labels = torch.randint(0, V, (n,), device=dev)
    labels[::3] = -100
    sl = torch.full((n,), 4, dtype=torch.int64, device=dev)
    ns = torch.tensor([5.0], device=dev)
    opcheck(torch.ops.mi355_touch.ce_fwd.default, (logits, labels, sl, ns, -100), test_utils=tests)
    a, b = r(256, 256, grad=True), r(384, 256, grad=True)
    opcheck(torch.ops.mi355_touch.gemm_tn.default, (a, b, None), test_utils=tests)
    # numerics of the registered autograd formula of the GEMM against torch
    out = torch.ops.mi355_touch.gemm_tn(a, b, None)
    dy = r(256, 384)
    da, db = torch.autograd.grad(out, [a, b], dy)
    ar, br = a.detach().float().requires_grad_(), b.detach().float().requires_grad_()
    dar, dbr = torch.autograd.grad(ar @ br.t(), [ar, br], dy.float())
    torch.testing.assert_close(da.float(), dar, rtol=2e-2, atol=0.3)
    torch.testing.assert_close(db.float(), dbr, rtol=2e-2, atol=0.3)
