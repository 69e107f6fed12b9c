"""Oracle: decoder / encoder maths in eager PyTorch (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference holds no model arithmetic of its own: it drives `transformers==4.51.3`
modules (pyproject.toml:13).  This file restates that arithmetic functionally over a
``state_dict`` with the HF parameter names, following

  RMSNorm      site-packages:transformers/models/llama/modeling_llama.py:53-67
  RoPE         ...:113-160  (+ llama3 scaling: modeling_rope_utils `_compute_llama3_parameters`)
  MLP          ...:174-176
  attention    ...:243-281 (eager_attention_forward), GQA repeat_kv
  layer        ...:306-324
  doc mask     transformers/integrations/flex_attention.py:190-201 as called from
               touchnet/models/kimi_audio/modeling_kimi_audio.py:582-585
  TouchAudio   touchnet/models/touch_audio/modeling_touch_audio.py:123-131
  Qwen2-Audio  touchnet/models/qwen2_audio/__init__.py:18-133 (tower), :186-229 (merge)

Pinned by tests/golden/tiny_llama_*.npz, touch_audio_*.npz, qwen2_audio_tower_*.npz,
rope_llama3.npz, docmask.npz (all produced by the reference stack, make_golden.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ primitives
def rms_norm(x, weight, eps):
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return weight * xf.to(dt)


def layer_norm(x, weight, bias, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def rope_inv_freq(head_dim, theta, scaling=None):
    """Default and llama3-scaled inverse frequencies, fp32 [head_dim/2]."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    if not scaling or scaling.get("rope_type", scaling.get("type", "default")) == "default":
        return inv
    assert scaling.get("rope_type", scaling.get("type")) == "llama3"
    factor = scaling["factor"]
    lo, hi = scaling["low_freq_factor"], scaling["high_freq_factor"]
    old = scaling["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv
    inv_l = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    medium = ~(wavelen < old / hi) * ~(wavelen > old / lo)
    return torch.where(medium, smoothed, inv_l)


def rope_cos_sin(position_ids, inv_freq, dtype):
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :].float()
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """q [B,Nh,T,D], k [B,Nkv,T,D], cos/sin [B,T,D] (half-split convention)."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def doc_causal_allow(doc_ids):
    """allow[b,q,kv] = (q >= kv) & (doc[b,q] > 0) & (doc[b,q] == doc[b,kv])."""
    T = doc_ids.shape[1]
    idx = torch.arange(T, device=doc_ids.device)
    return ((idx[:, None] >= idx[None, :])[None]
            & (doc_ids[:, :, None] > 0)
            & (doc_ids[:, :, None] == doc_ids[:, None, :]))


def attention(q, k, v, allow, scale, zero_masked_rows=True):
    """q [B,Nh,T,D]; k,v [B,Nkv,T,D]; allow bool [B,T,T] or None (= plain causal).
    Returns [B,T,Nh,D].  Fully-masked query rows give 0 (flex semantics) when
    ``zero_masked_rows``; eager HF would give a uniform average there."""
    B, Nh, T, D = q.shape
    g = Nh // k.shape[1]
    k = k.repeat_interleave(g, dim=1)
    v = v.repeat_interleave(g, dim=1)
    if allow is None:
        idx = torch.arange(T, device=q.device)
        allow = (idx[:, None] >= idx[None, :])[None].expand(B, T, T)
    s = torch.matmul(q, k.transpose(2, 3)) * scale
    s = s.masked_fill(~allow[:, None], torch.finfo(s.dtype).min)
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    if zero_masked_rows:
        p = p * allow.any(-1)[:, None, :, None].to(p.dtype)
    return torch.matmul(p, v).transpose(1, 2).contiguous()


def swiglu(gate, up):
    return F.silu(gate) * up


# ------------------------------------------------------------------ Llama / Qwen2 decoder
def decoder_layer(sd, pfx, cfg, h, cos, sin, allow):
    B, T, H = h.shape
    Nh, Nkv, D = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    eps = cfg["rms_norm_eps"]
    x = rms_norm(h, sd[pfx + "input_layernorm.weight"], eps)
    lin = lambda n, t: F.linear(t, sd[pfx + n + ".weight"], sd.get(pfx + n + ".bias"))
    q = lin("self_attn.q_proj", x).view(B, T, Nh, D).transpose(1, 2)
    k = lin("self_attn.k_proj", x).view(B, T, Nkv, D).transpose(1, 2)
    v = lin("self_attn.v_proj", x).view(B, T, Nkv, D).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    a = attention(q, k, v, allow, D ** -0.5).reshape(B, T, Nh * D)
    h = h + lin("self_attn.o_proj", a)
    x = rms_norm(h, sd[pfx + "post_attention_layernorm.weight"], eps)
    return h + lin("mlp.down_proj", swiglu(lin("mlp.gate_proj", x), lin("mlp.up_proj", x)))


def causal_lm_forward(sd, cfg, doc_ids, position_ids, input_ids=None, inputs_embeds=None, prefix=""):
    """LlamaForCausalLM / Qwen2ForCausalLM forward on a packed batch -> logits [B,T,V].
    ``cfg``: dict with HF config keys (+ "head_dim"); ``doc_ids`` = the packers'
    `attention_mask` (None = plain causal, the Qwen2-Audio training path)."""
    if inputs_embeds is None:
        inputs_embeds = F.embedding(input_ids, sd[prefix + "model.embed_tokens.weight"])
    h = inputs_embeds
    inv = rope_inv_freq(cfg["head_dim"], cfg["rope_theta"], cfg.get("rope_scaling")).to(h.device)
    cos, sin = rope_cos_sin(position_ids, inv, h.dtype)
    allow = doc_causal_allow(doc_ids) if doc_ids is not None else None
    for i in range(cfg["num_hidden_layers"]):
        h = decoder_layer(sd, f"{prefix}model.layers.{i}.", cfg, h, cos, sin, allow)
    h = rms_norm(h, sd[prefix + "model.norm.weight"], cfg["rms_norm_eps"])
    w = sd.get(prefix + "lm_head.weight", sd[prefix + "model.embed_tokens.weight"])
    return F.linear(h, w)


def touch_audio_forward(sd, cfg, input_ids, input_features, doc_ids, position_ids):
    """modeling_touch_audio.py:123-149: embeds = projector(features) + embed_tokens(ids)."""
    emb = F.embedding(input_ids, sd["language_model.model.embed_tokens.weight"])
    emb = F.linear(input_features.to(emb.dtype), sd["projector.weight"]) + emb
    return causal_lm_forward(sd, cfg, doc_ids, position_ids, inputs_embeds=emb, prefix="language_model.")


# ------------------------------------------------------------------ Qwen2-Audio tower
def tiled_positions(embed_pos, seq_len):
    """qwen2_audio/__init__.py:52-73 — slice, or tile the table beyond its length."""
    n = embed_pos.shape[0]
    if n >= seq_len:
        return embed_pos[:seq_len]
    reps, rem = divmod(seq_len, n)
    parts = [embed_pos] * reps + ([embed_pos[:rem]] if rem else [])
    return torch.cat(parts, dim=0)


def whisper_layer(sd, pfx, h, n_heads, allow):
    """Pre-LN Whisper encoder layer with attention forced causal
    (qwen2_audio/__init__.py:190-193).  h [B,T,C]."""
    B, T, C = h.shape
    D = C // n_heads
    lin = lambda n, t: F.linear(t, sd[pfx + n + ".weight"], sd.get(pfx + n + ".bias"))
    x = layer_norm(h, sd[pfx + "self_attn_layer_norm.weight"], sd[pfx + "self_attn_layer_norm.bias"])
    q = lin("self_attn.q_proj", x).view(B, T, n_heads, D).transpose(1, 2)
    k = lin("self_attn.k_proj", x).view(B, T, n_heads, D).transpose(1, 2)
    v = lin("self_attn.v_proj", x).view(B, T, n_heads, D).transpose(1, 2)
    a = attention(q, k, v, allow, D ** -0.5).reshape(B, T, C)
    h = h + lin("self_attn.out_proj", a)
    x = layer_norm(h, sd[pfx + "final_layer_norm.weight"], sd[pfx + "final_layer_norm.bias"])
    return h + lin("fc2", F.gelu(lin("fc1", x)))


def qwen2_audio_tower(sd, acfg, mel, prefix="audio_tower."):
    """mel [B, n_mels, Tm] -> [B, Tm//4, d_model]  (conv stem, +pos, N layers, avgpool, LN)."""
    g = lambda n: sd[prefix + n]
    x = F.gelu(F.conv1d(mel, g("conv1.weight"), g("conv1.bias"), padding=1))
    x = F.gelu(F.conv1d(x, g("conv2.weight"), g("conv2.bias"), stride=2, padding=1))
    x = x.permute(0, 2, 1)
    h = x + tiled_positions(g("embed_positions.weight"), x.shape[1])[None]
    for i in range(acfg["encoder_layers"]):
        h = whisper_layer(sd, f"{prefix}layers.{i}.", h, acfg["encoder_attention_heads"], None)
    h = F.avg_pool1d(h.permute(0, 2, 1), 2, 2).permute(0, 2, 1)
    return layer_norm(h, g("layer_norm.weight"), g("layer_norm.bias"))


# ------------------------------------------------------------------ Kimi-Audio decoder (config E)
def kimi_audio_forward(sd, cfg, audio_input_ids, text_input_ids, doc_ids, position_ids, with_mimo=True):
    """MoonshotKimiaForCausalLM.forward + MoonshotKimiaModel.forward restated
    (touchnet/models/kimi_audio/modeling_kimi_audio.py:1026-1068 and :486-537; the layers are Qwen2 decoder layers,
    `decoder_layer` above): inputs = embed(audio ids) + embed(text ids); after layer `kimia_mimo_transformer_from_layer_index`
    the hidden state is CLONED into the mimo branch (`:506-507`), which runs `kimia_mimo_layers` more layers and its own
    final norm; returns (text_logits, audio_logits).  PARITY UNPINNED against the reference's own module: it does not
    import under the installed transformers 5.x, so this follows the source by hand."""
    emb_w = sd["model.embed_tokens.weight"]
    h = F.embedding(audio_input_ids, emb_w)
    if text_input_ids is not None:
        h = h + F.embedding(text_input_ids, emb_w)
    inv = rope_inv_freq(cfg["head_dim"], cfg["rope_theta"], cfg.get("rope_scaling")).to(h.device)
    cos, sin = rope_cos_sin(position_ids, inv, h.dtype)
    allow = doc_causal_allow(doc_ids) if doc_ids is not None else None
    mimo = None
    for i in range(cfg["num_hidden_layers"]):
        h = decoder_layer(sd, f"model.layers.{i}.", cfg, h, cos, sin, allow)
        if i == cfg["kimia_mimo_transformer_from_layer_index"]:
            mimo = h.clone()
    h = rms_norm(h, sd["model.norm.weight"], cfg["rms_norm_eps"])
    text_logits = F.linear(h, sd["lm_head.weight"])
    audio_logits = None
    if with_mimo:
        for i in range(cfg["kimia_mimo_layers"]):
            mimo = decoder_layer(sd, f"model.mimo_layers.{i}.", cfg, mimo, cos, sin, allow)
        mimo = rms_norm(mimo, sd["model.mimo_norm.weight"], cfg["rms_norm_eps"])
        audio_logits = F.linear(mimo, sd["mimo_output.weight"])
    return text_logits, audio_logits
