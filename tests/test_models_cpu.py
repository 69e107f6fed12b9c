"""Host logic of the product models on CPU: module wiring, HF parameter names, fused-residual scheduling.
The ops are the ORACLE's (explicitly injected — the product itself has no CPU path); what is under test
is everything around them, against the reference-generated goldens."""
import numpy as np
import pytest
import torch

import oracle.ops as oops
from touchnet_amd.loss.cross_entropy import cross_entropy_loss
from touchnet_amd.models.backend import use_ops
from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM, get_num_flop_per_token, get_num_params
from touchnet_amd.models.qwen2_audio import AudioEncoderConfig
from touchnet_amd.models.qwen2_audio.modeling_qwen2_audio import Qwen2AudioEncoder
from touchnet_amd.models.touch_audio import TouchAudioConfig, TouchAudioForCausalLM
from touchnet_amd.utils.metrics import accuracy

TINY = dict(vocab_size=16, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=8,
            num_key_value_heads=4, head_dim=8, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=True,
            rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                              original_max_position_embeddings=64))


def _load(model, g, prefix="param/"):
    sd = {k[len(prefix):]: torch.tensor(g[k]) for k in g.files if k.startswith(prefix)}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("lm_head" in m for m in missing), missing          # tied head is absent from HF's named_parameters
    return sd


def _check(model, g, fwd):
    batch = {k[len("batch/"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch/")}
    with use_ops(oops):
        logits = fwd(batch).logits
        valid = batch["attention_mask"] > 0
        np.testing.assert_allclose(logits[valid].detach().numpy(), g["logits"][valid.numpy()], atol=2e-5)
        ps, pt = cross_entropy_loss(logits, batch["labels"], batch["sentence_lens"], int(batch["num_sentence"]))
        acc = accuracy(logits, batch["labels"])
        ps.backward()
    assert float(ps) == pytest.approx(float(g["loss_per_sample"]), abs=1e-5)
    assert float(pt) == pytest.approx(float(g["loss_per_token"]), abs=1e-5)
    assert 0.0 <= float(acc) <= 1.0
    for name, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g["grad/" + name], atol=2e-5, err_msg=name)


def test_tiny_llama_matches_reference_stack(golden):
    g = golden("tiny_llama.npz")
    model = PackedCausalLM(DecoderConfig.from_dict(TINY))
    _load(model, g)
    assert {n for n, _ in model.named_parameters()} == {k[len("param/"):] for k in g.files if k.startswith("param/")}
    _check(model, g, lambda b: model(input_ids=b["input_ids"], position_ids=b["position_ids"],
                                     attention_mask=b["attention_mask"]))


def test_touch_audio_matches_reference(golden):
    g = golden("touch_audio.npz")
    model = TouchAudioForCausalLM(TouchAudioConfig(text_config=DecoderConfig.from_dict(TINY), input_size=21))
    _load(model, g)
    _check(model, g, lambda b: model(input_ids=b["input_ids"], input_features=b["input_features"],
                                     position_ids=b["position_ids"], attention_mask=b["attention_mask"]))


def test_qwen2_audio_tower_matches_reference(golden):
    g = golden("qwen2_audio_tower.npz")
    tower = Qwen2AudioEncoder(AudioEncoderConfig(num_mel_bins=8, d_model=32, encoder_layers=2,
                                                 encoder_attention_heads=4, encoder_ffn_dim=64,
                                                 max_source_positions=10))
    sd = {k[len("param/"):]: torch.tensor(g[k]) for k in g.files if k.startswith("param/")}
    sd["embed_positions.weight"] = torch.tensor(g["embed_positions"])
    tower.load_state_dict(sd, strict=True)
    with use_ops(oops), torch.no_grad():
        out = tower(torch.tensor(g["mel"]))
    np.testing.assert_allclose(out.numpy(), g["out"], atol=2e-5)


def test_qwen2_audio_tower_on_kept_frames_only_equals_the_padded_schedule(golden):
    """forward_valid (the encoder layers on frames [0, 2 len_i) of every clip, packed with one document id per clip)
    must return exactly the rows the reference's schedule keeps: the padded forward (checked against the reference
    above) followed by `:202-205`'s compaction.  Exact because the tower's attention is forced causal."""
    g = golden("qwen2_audio_tower.npz")
    tower = Qwen2AudioEncoder(AudioEncoderConfig(num_mel_bins=8, d_model=32, encoder_layers=2,
                                                 encoder_attention_heads=4, encoder_ffn_dim=64,
                                                 max_source_positions=10))
    sd = {k[len("param/"):]: torch.tensor(g[k]) for k in g.files if k.startswith("param/")}
    sd["embed_positions.weight"] = torch.tensor(g["embed_positions"])
    tower.load_state_dict(sd, strict=True)
    mel = torch.tensor(g["mel"])
    gen = torch.Generator().manual_seed(3)
    mel = torch.cat([mel, torch.randn(3, *mel.shape[1:], generator=gen)])       # more clips than the golden has
    with use_ops(oops), torch.no_grad():
        full = tower(mel)                                                       # [n, Ta, C]
        n, Ta, _ = full.shape
        for lens in ([Ta] * n, [max(Ta - 1 - i, 0) for i in range(n)], [1] + [0] * (n - 2) + [Ta]):
            lens_t = torch.tensor(lens)
            if int(lens_t.sum()) == 0:
                continue
            got = tower.forward_valid(mel, lens_t, int(lens_t.sum()))
            want = torch.cat([full[i, :l] for i, l in enumerate(lens)])
            assert got.shape == want.shape
            np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-6)


@pytest.mark.parametrize("bound,T", [(40, 288), (7, 300)])
def test_last_layer_on_the_labelled_rows_only_is_the_same_training_step(monkeypatch, bound, T):
    """With `labelled_rows_max` the rows that carry a label are selected in front of the LAST layer's output projection
    (o_proj, residual, norm, MLP, final norm and lm_head then see those rows only).  Loss, accuracy and EVERY gradient
    must equal the full computation (fp32 oracle ops); a bound whose 256-rounding is below the real count (T = 300: 300
    labelled rows, bound 7 -> 256) must poison the loss AND all gradients with NaN instead of dropping labels."""
    import touchnet_amd.models.llama.modeling_llama as ml
    torch.manual_seed(0)
    cfg = DecoderConfig.from_dict(dict(TINY, num_hidden_layers=3))
    model = PackedCausalLM(cfg)
    model.post_init()
    B = 2
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 16, (B, T), generator=g)
    doc = torch.ones(B, T, dtype=torch.int64)
    doc[0, 20:] = 2
    doc[1, 40:] = 0
    pos = torch.arange(T).expand(B, T).clone()
    labels = torch.full((B, T), -100)
    sl = torch.ones(B, T, dtype=torch.int64)
    spans = ((0, 12, 20), (0, 39, 48), (1, 30, 36)) if T == 288 else ((0, 21, 171), (1, 0, 150))
    for b, s, e in spans:
        labels[b, s:e] = torch.randint(1, 16, (e - s,), generator=g)
        sl[b, s:e] = e - s
    n_lab = int((labels != -100).sum())
    assert n_lab == (23 if T == 288 else 300)
    kw = dict(input_ids=ids, position_ids=pos, attention_mask=doc, labels=labels, sentence_lens=sl, num_sentence=3)

    taken = []
    inner = PackedCausalLM._forward_labelled_rows
    monkeypatch.setattr(PackedCausalLM, "_forward_labelled_rows", lambda self, *a, **k: (taken.append(1), inner(self, *a, **k))[1])

    def run(flag):
        monkeypatch.setattr(ml, "LAST_LAYER_LABELLED_ROWS", flag)
        model.zero_grad()
        with use_ops(oops):
            out = model(**kw, labelled_rows_max=bound)
            out.loss.backward()
        return out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    full, gfull = run(False)
    assert not taken
    rows, grows = run(True)
    assert len(taken) == 1
    if (bound + 255) // 256 * 256 < n_lab:
        assert torch.isnan(rows.loss) and torch.isnan(rows.loss_per_token) and torch.isnan(rows.acc)
        # ... and the poison reaches EVERY gradient, so that the optimizer's non-finite check skips the step (a NaN
        # loss with zero / truncated gradients would still apply weight decay and stale momentum)
        assert len(grows) > 20 and all(bool(torch.isnan(g).any()) for g in grows.values())
        return                      # (the oracle's fused CE computes every row: no bound, nothing to poison)
    assert float(rows.loss) == pytest.approx(float(full.loss), rel=1e-6)
    assert float(rows.loss_per_token) == pytest.approx(float(full.loss_per_token), rel=1e-6)
    assert float(rows.acc) == float(full.acc)
    assert gfull.keys() == grows.keys() and len(gfull) > 20
    for n in gfull:
        torch.testing.assert_close(grows[n], gfull[n], rtol=1e-5, atol=1e-7, msg=lambda m: f"{n}: {m}")


def test_meta_device_construction_and_counts():
    cfg = DecoderConfig.from_dict(TINY)
    with torch.device("meta"):
        m = PackedCausalLM(cfg)
    assert all(p.is_meta for p in m.parameters())
    n, ne = get_num_params(m), get_num_params(m, exclude_embedding=True)
    assert n - ne == 16 * 64 and get_num_flop_per_token(ne, cfg, 32) == 6 * ne + 12 * 2 * 8 * 8 * 32


def test_product_ops_refuse_cpu_tensors():
    """No silent fallback: the HIP wrappers must raise on CPU tensors."""
    import touchnet_amd.functional as F
    from touchnet_amd._C import KernelError
    # (library ops: the dispatcher itself refuses — "no kernel for the CPU backend" — a NotImplementedError)
    with pytest.raises((KernelError, ImportError, NotImplementedError)):
        F.rms_norm(torch.randn(4, 64), torch.ones(64), 1e-5)
    w = torch.randn(64, 64)
    with pytest.raises((KernelError, ImportError)):
        F.linear_group(torch.randn(8, 64), [(w, None)])
    with pytest.raises((KernelError, ImportError)):
        F.swiglu_mlp(torch.randn(8, 64), w, w, w)
    with pytest.raises((KernelError, ImportError, RuntimeError)):
        F.bestrq_tokenize(torch.randn(8, 16), torch.randn(16, 8), torch.randn(32, 8))


def _hf_packed_case():
    B, T = 2, 48
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, 97, (B, T), generator=g)
    docs = torch.tensor([[1] * 10 + [2] * 20 + [3] * 15 + [0] * 3, [1] * 30 + [2] * 18])
    pos = torch.cat([torch.cat([torch.arange(n) for n in lens]) for lens in ([10, 20, 15, 1, 1, 1], [30, 18])]).view(B, T)
    q = torch.arange(T)
    allow = (docs[:, :, None] == docs[:, None, :]) & (docs[:, :, None] > 0) & (q[None, None, :] <= q[None, :, None])
    return ids, docs, pos, allow


def test_hf_attention_interface_drives_transformers_llama():
    """SURVEY §8b hook 2: transformers' own LlamaForCausalLM with `attn_implementation="mi355_packed"` (our
    attention-function adapter, here on the oracle backend) == the same model in eager mode with the explicit 4-D
    document mask, with the document ids taken (a) from restarting position_ids, (b) from `document_ids=`, (c) from
    an integer 2-D attention_mask handed straight to the function."""
    from transformers import LlamaConfig, LlamaForCausalLM

    import oracle.ops as oops
    from touchnet_amd.integrations import hf_attention
    from touchnet_amd.models.backend import use_ops
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2,
                      num_hidden_layers=2, vocab_size=97, head_dim=16, max_position_embeddings=256)
    name = hf_attention.register()
    m = LlamaForCausalLM(cfg).eval()
    ids, docs, pos, allow = _hf_packed_case()
    bias = torch.zeros(ids.shape[0], 1, ids.shape[1], ids.shape[1]).masked_fill(~allow[:, None], torch.finfo(torch.float32).min)
    m.config._attn_implementation = "eager"
    with torch.no_grad():
        ref = m(input_ids=ids, position_ids=pos, attention_mask=bias).logits
    m.config._attn_implementation = name
    valid = docs > 0
    with use_ops(oops), torch.no_grad():
        a = m(input_ids=ids, position_ids=pos).logits
        b = m(input_ids=ids, position_ids=pos, document_ids=docs).logits
    assert float((a - ref)[valid].abs().max()) < 1e-5 and float((b - ref)[valid].abs().max()) < 1e-5
    # (c) + the layout contract of the function itself
    qh, kh, vh = torch.randn(2, 4, 48, 16), torch.randn(2, 2, 48, 16), torch.randn(2, 2, 48, 16)
    with use_ops(oops):
        out, w = hf_attention.packed_attention_forward(None, qh, kh, vh, attention_mask=docs, scaling=0.25)
    assert w is None and out.shape == (2, 48, 4, 16) and out.is_contiguous()
    assert torch.equal(hf_attention.documents_from_positions(pos)[0, :45], docs[0, :45].to(torch.int32))
    # the tile metadata is built ONCE per forward and shared by the layers, also for TouchNet's int64 document-id mask
    # (the key is the caller's tensor, not the int32 copy made per call); an in-place edit of the mask is seen
    docs64 = docs.to(torch.int64).clone()
    with use_ops(oops):
        before = hf_attention._cache.get("builds", 0)
        for _ in range(3):
            hf_attention.packed_attention_forward(None, qh, kh, vh, attention_mask=docs64, scaling=0.25)
        assert hf_attention._cache["builds"] == before + 1
        docs64[0, 0] = 7
        hf_attention.packed_attention_forward(None, qh, kh, vh, attention_mask=docs64, scaling=0.25)
        assert hf_attention._cache["builds"] == before + 2
    with pytest.raises(NotImplementedError):
        hf_attention.packed_attention_forward(None, qh, kh, vh, dropout=0.1)


def test_hf_module_swap_patch_keeps_transformers_llama_outputs():
    """SURVEY §8b hook 1: apply_mi355_kernels_to_llama() (RMSNorm + MLP + attention interface) on transformers'
    LlamaForCausalLM — oracle backend — gives the logits and gradients of the unpatched eager model; undo() restores."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama import modeling_llama as hf

    import oracle.ops as oops
    from touchnet_amd.integrations import hf_attention, hf_patch
    from touchnet_amd.models.backend import use_ops
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2,
                      num_hidden_layers=2, vocab_size=97, head_dim=16, max_position_embeddings=256)
    m = LlamaForCausalLM(cfg)
    ids, docs, pos, allow = _hf_packed_case()
    bias = torch.zeros(ids.shape[0], 1, ids.shape[1], ids.shape[1]).masked_fill(~allow[:, None], torch.finfo(torch.float32).min)
    valid = (docs > 0)[..., None].float()
    m.config._attn_implementation = "eager"
    ref = m(input_ids=ids, position_ids=pos, attention_mask=bias).logits
    (ref * valid).square().mean().backward()
    gref = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    orig_norm, orig_mlp = hf.LlamaRMSNorm.forward, hf.LlamaMLP.forward
    try:
        hf_patch.apply_mi355_kernels_to_llama()
        hf_patch.apply_mi355_kernels_to_llama()                     # idempotent
        assert hf.LlamaRMSNorm.forward is not orig_norm and hf.LlamaMLP.forward is not orig_mlp
        m.config._attn_implementation = hf_attention.NAME
        with use_ops(oops):
            got = m(input_ids=ids, position_ids=pos, document_ids=docs).logits
            (got * valid).square().mean().backward()
        assert float(((got - ref) * valid).abs().max()) < 1e-4
        for n, p in m.named_parameters():
            assert float((p.grad - gref[n]).abs().max()) < 1e-4 * max(1.0, float(gref[n].abs().max())), n
    finally:
        hf_patch.undo()
    assert hf.LlamaRMSNorm.forward is orig_norm and hf.LlamaMLP.forward is orig_mlp


def test_hf_module_swap_patch_qwen2():
    """Same as above for transformers' Qwen2ForCausalLM (q/k/v biases stay HF's nn.Linear; RMSNorm, MLP, attention swapped)."""
    from transformers import Qwen2Config, Qwen2ForCausalLM

    import oracle.ops as oops
    from touchnet_amd.integrations import hf_attention, hf_patch
    from touchnet_amd.models.backend import use_ops
    torch.manual_seed(1)
    cfg = Qwen2Config(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2,
                      num_hidden_layers=2, vocab_size=97, max_position_embeddings=256, use_sliding_window=False)
    m = Qwen2ForCausalLM(cfg).eval()
    ids, docs, pos, allow = _hf_packed_case()
    bias = torch.zeros(ids.shape[0], 1, ids.shape[1], ids.shape[1]).masked_fill(~allow[:, None], torch.finfo(torch.float32).min)
    m.config._attn_implementation = "eager"
    with torch.no_grad():
        ref = m(input_ids=ids, position_ids=pos, attention_mask=bias).logits
    try:
        hf_patch.apply_mi355_kernels_to_qwen2()
        m.config._attn_implementation = hf_attention.NAME
        with use_ops(oops), torch.no_grad():
            got = m(input_ids=ids, position_ids=pos, document_ids=docs).logits
    finally:
        hf_patch.undo()
    assert float((got - ref)[docs > 0].abs().max()) < 1e-4


def test_liger_branch_shift_labels_returns_token_mean_loss():
    """Hook 3, the reference's liger branch (train.py:434-445): called with `shift_labels` only, the model returns
    `.loss` = mean CE over the labelled tokens (and `.logits = None`), differentiable, equal to F.cross_entropy on the
    logits the same model returns without labels."""
    import oracle.ops as oops
    from touchnet_amd.models.backend import use_ops
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    torch.manual_seed(0)
    cfg = DecoderConfig.from_dict(dict(model_type="llama", hidden_size=64, intermediate_size=128, num_attention_heads=4,
                                       num_hidden_layers=2, num_key_value_heads=2, head_dim=16, vocab_size=97,
                                       tie_word_embeddings=False, rope_theta=10000.0, initializer_range=0.08))
    m = PackedCausalLM(cfg)
    m.post_init()
    ids, docs, pos, _ = _hf_packed_case()
    labels = ids.clone()
    labels[docs == 0] = -100
    labels[:, ::4] = -100
    with use_ops(oops):
        out = m(input_ids=ids, position_ids=pos, attention_mask=docs, shift_labels=labels)
        assert out.logits is None and out.loss.requires_grad
        logits = m(input_ids=ids, position_ids=pos, attention_mask=docs).logits
    want = torch.nn.functional.cross_entropy(logits.reshape(-1, 97).float(), labels.reshape(-1), ignore_index=-100)
    assert float(out.loss) == pytest.approx(float(want), rel=1e-5)
    out.loss.backward()
    assert m.lm_head.weight.grad is not None and torch.isfinite(m.lm_head.weight.grad).all()


def test_kimi_audio_decoder_wiring_equals_oracle_restatement():
    """Config E groundwork: the Kimi-Audio decoder (audio + text embedding sum, Qwen2 stack with q/k/v bias, mimo branch
    tapped after layer `kimia_mimo_transformer_from_layer_index`, two heads) on the oracle op set == the restatement of
    MoonshotKimiaModel / MoonshotKimiaForCausalLM (oracle/nn.py::kimi_audio_forward, by hand from
    modeling_kimi_audio.py:486-537, 1026-1068 — the reference module itself cannot be imported here), logits of both
    heads and the flop / parameter formulas of kimi_audio/__init__.py:63-93."""
    from oracle import nn as onn
    from touchnet_amd.models.kimi_audio import (KimiAudioConfig, KimiAudioPackedForCausalLM, get_num_flop_per_token,
                                                get_num_params)
    kw = dict(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=4, num_attention_heads=4,
              num_key_value_heads=2, head_dim=16, rms_norm_eps=1e-6, rope_theta=1e6, kimia_mimo_layers=2,
              kimia_mimo_transformer_from_layer_index=1)
    cfg = KimiAudioConfig(**kw)
    torch.manual_seed(0)
    m = KimiAudioPackedForCausalLM(cfg)
    m.post_init()
    for n, p in m.named_parameters():
        if n.endswith("bias"):
            torch.nn.init.normal_(p, std=0.05)
    B, T = 2, 48
    a, t = torch.randint(0, 64, (B, T)), torch.randint(0, 64, (B, T))
    doc = torch.cat([torch.ones(B, 20), 2 * torch.ones(B, 20), torch.zeros(B, 8)], 1).long()
    pos = torch.cat([torch.arange(20), torch.arange(20), torch.zeros(8, dtype=torch.long)]).repeat(B, 1)
    with use_ops(oops), torch.no_grad():
        out = m(text_input_ids=t, audio_input_ids=a, attention_mask=doc, position_ids=pos, compute_audio_logits=True)
        only_text = m(text_input_ids=t, audio_input_ids=a, attention_mask=doc, position_ids=pos)
    tl, al = onn.kimi_audio_forward(dict(m.state_dict()), kw, a, t, doc, pos)
    v = doc > 0
    assert float((out.logits - tl)[v].abs().max()) < 2e-5 and float((out.audio_logits - al)[v].abs().max()) < 2e-5
    assert only_text.audio_logits is None and torch.equal(only_text.logits, out.logits)
    n_wo = get_num_params(m, exclude_embedding=True)
    assert get_num_params(m) - n_wo == 64 * 64
    assert get_num_flop_per_token(n_wo, cfg, 48) == 6 * n_wo + 12 * (4 + 2) * 4 * 16 * 48
    # the text-head step does not run the mimo branch: its layers, norm and head are not credited
    executed = sum(p.numel() for n, p in m.named_parameters()
                   if not n.startswith(("model.mimo_layers.", "model.mimo_norm.", "mimo_output.", "model.embed_tokens.")))
    assert get_num_flop_per_token(n_wo, cfg, 48, with_mimo=False) == 6 * executed + 12 * 4 * 4 * 16 * 48
    assert any("q_proj.bias" in n for n, _ in m.named_parameters())          # Qwen2DecoderLayer: biased q/k/v


def test_kimi_audio_decoder_matches_the_reference_module(golden):
    """tests/golden/kimi_decoder.npz was produced by RUNNING the reference's MoonshotKimiaModel
    (modeling_kimi_audio.py:347-556) — make_golden.py::kimi_decoder_case, import recipe in _ref_import.py — on a packed
    two-document batch with the mimo branch: the product module (on the oracle op set) and the hand restatement
    oracle/nn.py::kimi_audio_forward both reproduce its text and audio logits, the reference CE loss and the gradients of
    every parameter the text-head loss reaches."""
    import ast

    from oracle import nn as onn
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    g = golden("kimi_decoder.npz")
    kw = ast.literal_eval(str(g["config_json"]))
    kw["head_dim"] = kw["hidden_size"] // kw["num_attention_heads"]
    m = KimiAudioPackedForCausalLM(KimiAudioConfig(**{k: v for k, v in kw.items() if k != "initializer_range"}))
    sd = {k[len("param/"):]: torch.tensor(g[k]) for k in g.files if k.startswith("param/")}
    assert set(sd) == {n for n, _ in m.named_parameters()}                    # same parameter names as the reference
    m.load_state_dict(sd, strict=True)
    b = {k[len("batch/"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch/")}
    valid = b["attention_mask"] > 0
    with use_ops(oops):
        out = m(text_input_ids=b["text_input_ids"], audio_input_ids=b["audio_input_ids"],
                attention_mask=b["attention_mask"], position_ids=b["position_ids"], compute_audio_logits=True)
        ps, pt = cross_entropy_loss(out.logits, b["labels"], b["sentence_lens"], 4)
        ps.backward()
    np.testing.assert_allclose(out.logits[valid].detach().numpy(), g["text_logits"][valid.numpy()], atol=3e-5)
    np.testing.assert_allclose(out.audio_logits[valid].detach().numpy(), g["audio_logits"][valid.numpy()], atol=3e-5)
    assert float(ps) == pytest.approx(float(g["loss_per_sample"]), abs=1e-5)
    assert float(pt) == pytest.approx(float(g["loss_per_token"]), abs=1e-5)
    checked = 0
    for n, p in m.named_parameters():
        if "grad/" + n in g.files:
            np.testing.assert_allclose(p.grad.numpy(), g["grad/" + n], atol=3e-5, err_msg=n)
            checked += 1
        else:
            assert "mimo" in n and (p.grad is None or float(p.grad.abs().max()) == 0.0), n
    assert checked == 4 * 12 + 1 + 1 + 1          # 4 layers x (7 weights + 3 biases + 2 norms), embedding, final norm, lm_head
    tl, al = onn.kimi_audio_forward(sd, kw, b["audio_input_ids"], b["text_input_ids"], b["attention_mask"], b["position_ids"])
    np.testing.assert_allclose(tl[valid].numpy(), g["text_logits"][valid.numpy()], atol=3e-5)
    np.testing.assert_allclose(al[valid].numpy(), g["audio_logits"][valid.numpy()], atol=3e-5)


@pytest.mark.parametrize("labelled", [None, 64])
@pytest.mark.parametrize("bound", ["exact", "too_small"])
def test_padding_slots_are_dropped_from_the_decoders_row_work_without_changing_the_step(monkeypatch, labelled, bound):
    """`valid_rows_max` (the packers' count of non-pad slots): the decoder gathers the non-pad positions of the packed batch
    into ONE row (document ids made unique across batch rows, count rounded up to 256 with real padding slots) and runs
    every layer on those rows only.  Loss, accuracy and EVERY gradient equal the full computation (fp32 oracle ops) — with
    and without the labelled-rows shortcut of the last layer; a bound below the real count poisons loss and gradients."""
    import touchnet_amd.models.llama.modeling_llama as ml
    torch.manual_seed(0)
    cfg = DecoderConfig.from_dict(dict(TINY, num_hidden_layers=3))
    model = PackedCausalLM(cfg)
    model.post_init()
    B, T = 3, 512
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, 16, (B, T), generator=g)
    doc = torch.zeros(B, T, dtype=torch.int64)
    pos = torch.zeros(B, T, dtype=torch.int64)
    labels = torch.full((B, T), -100)
    sl = torch.ones(B, T, dtype=torch.int64)
    # documents of uneven length; every batch row ends in padding, row 2 is mostly padding; SAME ids in different rows
    layout = {0: [(1, 0, 150), (2, 150, 330)], 1: [(1, 0, 90), (2, 90, 95), (3, 95, 400)], 2: [(1, 0, 37)]}
    n_sent = 0
    for b, docs in layout.items():
        for d, s, e in docs:
            doc[b, s:e] = d
            pos[b, s:e] = torch.arange(e - s)
            k = max(e - 11, s)
            labels[b, k:e] = torch.randint(1, 16, (e - k,), generator=g)
            sl[b, k:e] = e - k
            n_sent += 1
    n_valid = int((doc > 0).sum())
    assert n_valid == 330 + 400 + 37 and (n_valid + 255) // 256 * 256 + 256 <= B * T
    kw = dict(input_ids=ids, position_ids=pos, attention_mask=doc, labels=labels, sentence_lens=sl, num_sentence=n_sent)
    if labelled is not None:
        kw["labelled_rows_max"] = int((labels != -100).sum())
    taken = []
    inner = ml.DecoderModel._drop_pad_rows
    monkeypatch.setattr(ml.DecoderModel, "_drop_pad_rows",
                        staticmethod(lambda *a: (lambda r: (taken.append(r is not None), r)[1])(inner(*a))))

    def run(vmax):
        model.zero_grad()
        with use_ops(oops):
            out = model(**kw, valid_rows_max=vmax)
            out.loss.backward()
        return out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    full, gfull = run(None)
    assert not taken
    got, ggot = run(n_valid if bound == "exact" else 500)          # 500 -> 512 rows < 767 real tokens
    assert taken == [True]
    if bound != "exact":
        assert torch.isnan(got.loss) and all(bool(torch.isnan(v).any()) for v in ggot.values())
        return
    assert float(got.loss) == pytest.approx(float(full.loss), rel=1e-6)
    assert float(got.acc) == float(full.acc)
    assert gfull.keys() == ggot.keys() and len(gfull) > 20
    for n in gfull:
        torch.testing.assert_close(ggot[n], gfull[n], rtol=1e-5, atol=1e-7, msg=lambda m: f"{n}: {m}")
    # a caller that asks for logits gets every position computed (no compaction), whatever the batch carries
    taken.clear()
    with use_ops(oops), torch.no_grad():
        lg = model(input_ids=ids, position_ids=pos, attention_mask=doc, valid_rows_max=n_valid).logits
    assert lg.shape == (B, T, cfg.vocab_size) and not taken
    # nothing to save (bound within one tile row of B*T): the batch is left as it is
    with use_ops(oops):
        model(**kw, valid_rows_max=B * T - 100)
    assert taken == [False]


def test_op_level_ac_is_a_noop_under_sequence_parallelism_and_mismatched_norm_sources_fall_back():
    """selective AC option "op" marks decoder blocks so that their GEMM nodes keep the residual stream
    instead of the norm output.  Under tensor-parallel sequence parallelism the attention / MLP wrappers gather x to
    the full sequence BEHIND the norm: the local residual does not describe the GEMM input.  apply_ac must leave such
    a model unmarked (the reference's TP + "op" configuration is valid: touchnet/models/helper_func.py:39-96), and the
    two consumers of a norm source drop one that does not describe their input instead of raising."""
    import types
    import warnings

    import touchnet_amd.functional as F
    from touchnet_amd.models.parallelize import apply_ac

    cfg = DecoderConfig.from_dict(dict(model_type="llama", hidden_size=64, intermediate_size=128, num_attention_heads=4,
                                       num_hidden_layers=2, num_key_value_heads=2, head_dim=16, vocab_size=128))
    job = types.SimpleNamespace(training_activation_checkpoint_mode="selective",
                                training_activation_checkpoint_selective_ac_option="op")
    plain = PackedCausalLM(cfg)
    apply_ac(plain, job)
    assert all(getattr(b, "_tn_recompute_rows", False) for b in plain.model.layers)
    sp = PackedCausalLM(cfg)
    sp.model._tn_sp = object()                         # what models/tensor_parallel.apply_tp leaves on a decoder stack
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        apply_ac(sp, job)
    assert not any(getattr(b, "_tn_recompute_rows", False) for b in sp.model.layers)
    assert any("sequence parallelism" in str(x.message) for x in w)
    # a norm source over T/tp local rows against the gathered [B, T, H] input: not a description of x
    h_local, x_full = torch.zeros(1, 8, 64), torch.zeros(1, 16, 64)
    src = F.norm_source(h_local, torch.ones(64), 1e-6)
    assert not F._norm_src_describes(src, x_full)
    assert F._norm_src_describes(F.norm_source(x_full, torch.ones(64), 1e-6), x_full)
    assert not F._norm_src_describes(None, x_full)
    assert not F._norm_src_describes(F.norm_source(x_full.double(), torch.ones(64), 1e-6), x_full)


def test_rope_frequencies_stay_float32_when_the_model_is_cast():
    """`model.to(torch.bfloat16)` — the single-GPU Trainer's mixed-precision cast — must not round inv_freq to 8 bits (the
    reference keeps it float32: touchnet/models/llama/__init__.py:19-36 re-derives it on the init device in fp32).  Found by
    tests/test_full_size_parity_gpu.py: at position 700 the fastest pairs were off by radians."""
    cfg = DecoderConfig.from_dict(dict(model_type="qwen2", hidden_size=64, intermediate_size=128, num_attention_heads=4,
                                       num_hidden_layers=1, num_key_value_heads=2, head_dim=16, vocab_size=128,
                                       rope_theta=1000000.0))
    m = PackedCausalLM(cfg)
    ref = m.model.rotary_emb.inv_freq.clone()
    m.to(torch.bfloat16)
    assert m.lm_head.weight.dtype == torch.bfloat16
    assert m.model.rotary_emb.inv_freq.dtype == torch.float32 and torch.equal(m.model.rotary_emb.inv_freq, ref)
    m.to(torch.float64).to(torch.bfloat16)
    assert torch.equal(m.model.rotary_emb.inv_freq, ref)
    assert "model.rotary_emb.inv_freq" not in m.state_dict()            # (still a non-persistent buffer)
