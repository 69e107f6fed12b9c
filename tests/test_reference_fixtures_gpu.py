"""The device <-> reference loop WITHOUT the product's own wiring in between.

tests/golden/*_dev.npz hold weights, batch, logits, loss and gradients produced by RUNNING the reference's modules
(transformers' LlamaForCausalLM under the reference's batcher and loss, touchnet's TouchAudioForCausalLM, the Qwen2-Audio
tower driven by the reference's `forward_audio_tower`, touchnet's MoonshotKimiaModel) at the smallest widths the MI355X
kernels take (head_dim 64), in float32 on bf16-rounded weights — tests/golden/make_golden.py::*_dev_case.  Here the fixture
weights go INTO THE PRODUCT MODEL ON THE MI355X and its outputs are held to the fixture's: loss within north_star's 1e-3
relative, logits on valid rows and every gradient at bf16 tolerance (observed values are printed; the bounds are observed +
margin).  The CPU twin (`-m "not gpu"`) runs the same comparison on the oracle op set at float32 tolerance, which pins
the loaders and the fixture itself."""
import ast

import numpy as np
import pytest
import torch

import oracle.ops as oops
from touchnet_amd.loss.cross_entropy import cross_entropy_loss
from touchnet_amd.models.backend import use_ops

DEV = "cuda"
LOSS_REL = 1e-3


def _bf16(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def _state(g, prefix="param/"):
    return {k[len(prefix):]: _bf16(g[k]).float() for k in g.files if k.startswith(prefix)}


def _batch(g):
    return {k[len("batch/"):]: torch.tensor(g[k]) for k in g.files if k.startswith("batch/")}


def _llama(g):
    from touchnet_amd.models.llama import DecoderConfig, PackedCausalLM
    cfg = DecoderConfig.from_dict(dict(ast.literal_eval(str(g["config_json"])), model_type="llama"))
    m = PackedCausalLM(cfg)
    missing, unexpected = m.load_state_dict(_state(g), strict=False)
    assert not unexpected and all("lm_head" in k for k in missing)
    return m, lambda mod, b: mod(input_ids=b["input_ids"], position_ids=b["position_ids"], attention_mask=b["attention_mask"])


def _touch_audio(g):
    from touchnet_amd.models.llama import DecoderConfig
    from touchnet_amd.models.touch_audio import TouchAudioConfig, TouchAudioForCausalLM
    text = dict(ast.literal_eval(str(np.load(g.fid.name.replace("touch_audio_dev", "tiny_llama_dev"))["config_json"])),
                model_type="llama")
    m = TouchAudioForCausalLM(TouchAudioConfig(text_config=DecoderConfig.from_dict(text), input_size=int(g["input_size"])))
    missing, unexpected = m.load_state_dict(_state(g), strict=False)
    assert not unexpected and all("lm_head" in k for k in missing)
    return m, lambda mod, b: mod(input_ids=b["input_ids"], input_features=b["input_features"],
                                 position_ids=b["position_ids"], attention_mask=b["attention_mask"])


def _qwen2_audio(g):
    """The WHOLE Qwen2-Audio model on a batch in the reference's own unpacked format: `feature_attention_mask`, 0/1
    `attention_mask`, no position ids, no audio index tensors — the keys processing_qwen2_audio.py:119-147 yields."""
    from touchnet_amd.models.qwen2_audio import Qwen2AudioConfig, Qwen2AudioPackedForConditionalGeneration
    d = ast.literal_eval(str(g["config_json"]))
    cfg = Qwen2AudioConfig.from_dict({"audio_config": d["audio_config"], "audio_token_index": d["audio_token_index"],
                                      "text_config": dict(d["text_config"], model_type="qwen2")})
    m = Qwen2AudioPackedForConditionalGeneration(cfg)
    missing, unexpected = m.load_state_dict(_state(g), strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m, lambda mod, b: mod(input_ids=b["input_ids"], input_features=b["input_features"],
                                 attention_mask=b["attention_mask"], feature_attention_mask=b["feature_attention_mask"],
                                 shift_labels=b["labels"])


def _compare(m, g, fwd, batch, device, logit_tol, logit_mean_tol, grad_tol, loss_tol):
    data = {k: v.to(device) for k, v in batch.items() if k != "num_sentence"}
    if "input_features" in data and device != "cpu":
        data["input_features"] = data["input_features"].to(torch.bfloat16).float()
    pred = fwd(m, data)
    ns = int(batch["num_sentence"])
    ps, pt = cross_entropy_loss(pred.logits, data["labels"], data["sentence_lens"], ns)
    ps.backward()
    valid = (batch["attention_mask"] > 0).numpy()
    ref = g["logits"]
    got = pred.logits.detach().float().cpu().numpy()
    scale = float(np.abs(ref[valid]).max())
    e_max = float(np.abs(got[valid] - ref[valid]).max()) / scale
    e_mean = float(np.abs(got[valid] - ref[valid]).mean()) / scale
    rel = abs(float(ps) - float(g["loss_per_sample"])) / abs(float(g["loss_per_sample"]))
    rel_t = abs(float(pt) - float(g["loss_per_token"])) / abs(float(g["loss_per_token"]))
    worst = []
    for n, p in m.named_parameters():
        key = "grad/" + n
        if key not in g.files:
            continue
        r = g[key].astype(np.float32)
        denom = max(float(np.abs(r).max()), 1e-6)
        worst.append((float(np.abs(p.grad.float().cpu().numpy() - r).max()) / denom, n))
    worst.sort(reverse=True)
    print(f"FIXTURE PARITY ({device}) loss rel {rel:.2e} / {rel_t:.2e}, logits max {e_max:.2e} mean {e_mean:.2e}, "
          f"worst grads {[(round(e, 4), n) for e, n in worst[:3]]}")
    assert rel < loss_tol and rel_t < loss_tol, (rel, rel_t)
    assert e_max < logit_tol and e_mean < logit_mean_tol, (e_max, e_mean)
    assert worst and worst[0][0] < grad_tol, worst[:5]


# ------------------------------------------------------------------------------------------------ CPU: pins the fixtures
@pytest.mark.parametrize("name,build", [("tiny_llama_dev.npz", _llama), ("touch_audio_dev.npz", _touch_audio),
                                        ("qwen2_audio_model_dev.npz", _qwen2_audio)])
def test_dev_fixture_on_the_oracle_ops(golden, name, build):
    g = golden(name)
    m, fwd = build(g)
    with use_ops(oops):
        # (gradients are stored as float16: 5e-4 of their scale; logits / losses are float32)
        _compare(m, g, fwd, _batch(g), "cpu", 2e-5, 2e-6, 1e-3, 1e-5)


def test_kimi_dev_fixture_on_the_oracle_ops(golden):
    _kimi(golden("kimi_decoder_dev.npz"), "cpu", 3e-5, 1e-3, 1e-5)


def test_tower_dev_fixture_on_the_oracle_ops(golden):
    _tower(golden("qwen2_audio_tower_dev.npz"), "cpu", 3e-5)


def test_kimi_audio_input_side_on_the_oracle_ops(golden):
    _kimi_input(golden("kimi_audio_input.npz"), "cpu", 3e-5, 2e-3)


# ------------------------------------------------------------------------------------------------ MI355X
@pytest.mark.gpu
@pytest.mark.parametrize("name,build", [("tiny_llama_dev.npz", _llama), ("touch_audio_dev.npz", _touch_audio),
                                        ("qwen2_audio_model_dev.npz", _qwen2_audio)])
def test_dev_fixture_on_the_device(golden, name, build):
    g = golden(name)
    m, fwd = build(g)
    m = m.to(DEV).to(torch.bfloat16)
    # observed on the MI355X (profiles/r04b_*): loss 3e-5 .. 7e-5 rel, logits 2.7e-3 max / 4.3e-4 mean of their scale, worst
    # gradient 1.15 % of its own scale
    _compare(m, g, fwd, _batch(g), DEV, 1e-2, 1.5e-3, 3e-2, LOSS_REL)


@pytest.mark.gpu
def test_kimi_dev_fixture_on_the_device(golden):
    # observed: logits 2.0 % of their scale; gradients <= 2.6 % of their own scale.  The k_proj BIAS gradients are bounded on
    # the scale of the layer's q / k / v bias gradients (VERDICT r5 / profiles/r05k_*: a key bias shifts every score of a row
    # alike wherever RoPE rotates slowly, the column sum of dK cancels there and the reference vector is 5 x smaller than
    # its neighbours while the error stays the absolute bf16 floor every bias gradient has).
    _kimi(golden("kimi_decoder_dev.npz"), DEV, 3e-2, 3e-2, LOSS_REL)


@pytest.mark.gpu
def test_tower_dev_fixture_on_the_device(golden):
    _tower(golden("qwen2_audio_tower_dev.npz"), DEV, 2e-2)                       # observed 8.4e-3


def _kimi(g, device, logit_tol, grad_tol, loss_tol):
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    kw = ast.literal_eval(str(g["config_json"]))
    kw["head_dim"] = kw["hidden_size"] // kw["num_attention_heads"]
    m = KimiAudioPackedForCausalLM(KimiAudioConfig(**{k: v for k, v in kw.items() if k != "initializer_range"}))
    sd = _state(g)
    assert set(sd) == {n for n, _ in m.named_parameters()}
    m.load_state_dict(sd, strict=True)
    b = _batch(g)
    if device != "cpu":
        m = m.to(device).to(torch.bfloat16)
    d = {k: v.to(device) for k, v in b.items()}
    valid = (b["attention_mask"] > 0).numpy()

    def run():
        out = m(text_input_ids=d["text_input_ids"], audio_input_ids=d["audio_input_ids"], attention_mask=d["attention_mask"],
                position_ids=d["position_ids"], compute_audio_logits=True)
        ps, pt = cross_entropy_loss(out.logits, d["labels"], d["sentence_lens"], 4)
        ps.backward()
        return out, ps, pt
    if device == "cpu":
        with use_ops(oops):
            out, ps, pt = run()
    else:
        out, ps, pt = run()
    for got, ref in ((out.logits, g["text_logits"]), (out.audio_logits, g["audio_logits"])):
        got = got.detach().float().cpu().numpy()
        scale = float(np.abs(ref[valid]).max())
        assert float(np.abs(got[valid] - ref[valid]).max()) / scale < logit_tol
    assert abs(float(ps) - float(g["loss_per_sample"])) / abs(float(g["loss_per_sample"])) < loss_tol
    assert abs(float(pt) - float(g["loss_per_token"])) / abs(float(g["loss_per_token"])) < loss_tol
    worst = []
    for n, p in m.named_parameters():
        if "grad/" + n in g.files:
            r = g["grad/" + n].astype(np.float32)
            denom = float(np.abs(r).max())
            if n.endswith("k_proj.bias"):       # scale of the layer's three projection-bias gradients (see the caller)
                sib = [n.replace("k_proj", x) for x in ("q_proj", "k_proj", "v_proj")]
                denom = max(float(np.abs(g["grad/" + x].astype(np.float32)).max()) for x in sib if "grad/" + x in g.files)
            worst.append((float(np.abs(p.grad.float().cpu().numpy() - r).max()) / max(denom, 1e-6), n))
    worst.sort(reverse=True)
    print(f"FIXTURE PARITY kimi ({device}) worst grads {[(round(e, 4), n) for e, n in worst[:3]]}")
    assert len(worst) == 2 * 12 + 1 + 1 + 1 and worst[0][0] < grad_tol, worst[:5]


def _tower(g, device, tol):
    from touchnet_amd.models.qwen2_audio import AudioEncoderConfig
    from touchnet_amd.models.qwen2_audio.modeling_qwen2_audio import Qwen2AudioEncoder
    tower = Qwen2AudioEncoder(AudioEncoderConfig(**ast.literal_eval(str(g["config_json"]))))
    sd = _state(g)
    sd["embed_positions.weight"] = _bf16(g["embed_positions"]).float()
    tower.load_state_dict(sd, strict=True)
    mel = torch.tensor(g["mel"])
    with torch.no_grad():
        if device == "cpu":
            with use_ops(oops):
                out = tower(mel)
        else:
            out = tower.to(device).to(torch.bfloat16)(mel.to(device).to(torch.bfloat16))
    ref = g["out"]
    err = float(np.abs(out.float().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
    print(f"FIXTURE PARITY tower ({device}) max err {err:.2e} of scale")
    assert err < tol, err


@pytest.mark.gpu
def test_kimi_audio_input_side_on_the_device(golden):
    """speech encoder (bidirectional attention = two launches of the causal kernel + the diagonal correction), x4 stack +
    VQ adaptor, embedding sum x sqrt(2), scatter between the media markers — against what the reference's
    `prepare_audio_input_embs` returned, forward and every gradient."""
    _kimi_input(golden("kimi_audio_input.npz"), DEV, 3e-2, 5e-2)      # observed: 9.4e-3 of scale, worst gradient 1.7 %


def _kimi_input(g, device, tol, grad_tol):
    """tests/golden/kimi_audio_input.npz: MoonshotKimiaForCausalLM.prepare_audio_input_embs (modeling_kimi_audio.py:933-985)
    run from the reference's source (make_golden.py::kimi_audio_input_case)."""
    from touchnet_amd.models.kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM
    kw = ast.literal_eval(str(g["config_json"]))
    begin, end = (int(v) for v in g["markers"])
    enc = {k: kw[k] for k in ("num_mel_bins", "d_model", "encoder_layers", "encoder_attention_heads", "encoder_ffn_dim",
                              "max_source_positions")}
    cfg = KimiAudioConfig(vocab_size=kw["vocab_size"], hidden_size=kw["hidden_size"], intermediate_size=128,
                          num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, head_dim=64,
                          rms_norm_eps=kw["rms_norm_eps"], kimia_mimo_layers=1, kimia_mimo_transformer_from_layer_index=0,
                          kimia_token_offset=kw["kimia_token_offset"], kimia_media_begin=begin, kimia_media_end=end,
                          use_whisper_feature=True, kimia_adaptor_input_dim=kw["kimia_adaptor_input_dim"],
                          speech_encoder_config=enc)
    m = KimiAudioPackedForCausalLM(cfg)
    sd = _state(g)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if k.startswith(("speech_encoder.", "model.vq_adaptor.", "model.embed_tokens."))], missing
    a = torch.tensor(g["audio_input_ids"])
    feats, ids, readout = torch.tensor(g["whisper_input_features"]), torch.tensor(g["speech_tokenizer_ids"]), torch.tensor(g["readout"])
    if device != "cpu":
        m = m.to(device).to(torch.bfloat16)
        a, ids, readout = a.to(device), ids.to(device), readout.to(device)
        feats = feats.to(device).to(torch.bfloat16)

    def run():
        out = m.prepare_audio_input_embs(a, m.model.embed_tokens(a), feats, ids)
        (out.float() * readout).sum().backward()
        return out
    if device == "cpu":
        with use_ops(oops):
            out = run()
    else:
        out = run()
    ref = g["out"]
    err = float(np.abs(out.detach().float().cpu().numpy() - ref).max()) / float(np.abs(ref).max())
    worst = []
    for n, p in m.named_parameters():
        if "grad/" + n in g.files:
            r = g["grad/" + n].astype(np.float32)
            worst.append((float(np.abs(p.grad.float().cpu().numpy() - r).max()) / max(float(np.abs(r).max()), 1e-6), n))
    worst.sort(reverse=True)
    print(f"FIXTURE PARITY kimi input side ({device}) out {err:.2e} of scale, worst grads {[(round(e, 4), n) for e, n in worst[:3]]}")
    assert err < tol, err
    assert len(worst) >= 2 * 15 + 4 + 6 + 1 - 2 and worst[0][0] < grad_tol, worst[:5]
