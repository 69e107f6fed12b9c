"""Kimi-Audio-7B decoder on the MI355X kernels — BASELINE config E groundwork.

Reference: `MoonshotKimiaModel` / `MoonshotKimiaForCausalLM` (touchnet/models/kimi_audio/modeling_kimi_audio.py:347-556,
847-1081): a Qwen2 decoder stack (H = 3584, 28 query / 4 kv heads, D = 128, I = 18944, 28 layers, q/k/v bias,
rope theta 1e6, eps 1e-6) whose input is the SUM of an audio-token and a text-token embedding (`:1030-1035`), plus a
six-layer "mimo" branch that starts from the hidden state after layer `kimia_mimo_transformer_from_layer_index` (21,
`:506-507, 519-537`) and feeds a second head (`mimo_output`, audio logits).  TouchNet trains on the TEXT logits only
(`:1066-1081`: "currently only support ASR task, so we only return text_logits") — the mimo branch is executed by the
reference but receives no gradient; here it runs only on request (`compute_audio_logits=True`), the flop formula still
counts its layers like the reference's (`kimi_audio/__init__.py:63-80`).

In scope: the decoder (the packed hot path: document-masked attention, fused norms / RoPE / SwiGLU, fused lm_head + CE
over V = 168448 on the labelled rows).  Not in this round: the Whisper-large-v3 speech encoder, the GLM-4-voice VQ
tokenizer and the VQ adaptor that turn waveforms into the continuous part of the audio embeddings
(`prepare_audio_input_embs`, `:942-985`) — their output enters through `audio_input_embs`.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from ..llama.configuration import DecoderConfig
from ..llama.modeling_llama import DecoderLayer, RMSNorm, RotaryEmbedding
from ..backend import ops


@dataclass
class KimiAudioConfig(DecoderConfig):
    """examples/audio/sft/asr/wenetspeech/config/Kimi-Audio-7B.json (the decoder keys)."""
    kimia_mimo_layers: int = 6
    kimia_mimo_transformer_from_layer_index: int = 21
    kimia_token_offset: int = 152064
    kimia_media_begin: int = 151661
    kimia_media_end: int = 151663

    def __post_init__(self):
        self.model_type = "qwen2"            # Qwen2DecoderLayer: q/k/v carry a bias
        super().__post_init__()
        self.attention_bias = True


class KimiDecoderModel(nn.Module):
    def __init__(self, config: KimiAudioConfig):
        super().__init__()
        self.config = config
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.rotary_emb = RotaryEmbedding(config)
        self.mimo_layers = nn.ModuleList([DecoderLayer(config) for _ in range(config.kimia_mimo_layers)])
        self.mimo_norm = RMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, inputs_embeds, position_ids=None, attention_mask=None, with_mimo: bool = False):
        B, T, _ = inputs_embeds.shape
        if position_ids is None:
            position_ids = torch.arange(T, device=inputs_embeds.device).expand(B, T)
        cos, sin = self.rotary_emb(position_ids, inputs_embeds.dtype)
        mask = attention_mask
        if mask is None:
            mask = ops().causal_mask(B, T, inputs_embeds.device)
        elif isinstance(mask, torch.Tensor):
            mask = ops().build_packed_mask(mask)
        sp = getattr(self, "_tn_sp", None)                  # sequence parallelism (models/tensor_parallel.py)
        delta, residual = (inputs_embeds if sp is None else sp.scatter(inputs_embeds)), None
        tap = None
        for idx, layer in enumerate(self.layers):
            delta, residual = layer(delta, residual, cos, sin, mask)
            if with_mimo and idx == self.config.kimia_mimo_transformer_from_layer_index:
                tap = residual + delta               # the hidden state after this layer (`:506-507`)
        h, _ = self.norm(delta, residual)
        h = h if sp is None else sp.gather(h)
        mimo = None
        if with_mimo:
            d, r = tap, None
            for layer in self.mimo_layers:
                d, r = layer(d, r, cos, sin, mask)
            mimo, _ = self.mimo_norm(d, r)
            mimo = mimo if sp is None else sp.gather(mimo)
        return h, mimo


class KimiAudioPackedForCausalLM(nn.Module):
    base_model_prefix = "model"
    config_class = KimiAudioConfig

    def __init__(self, config: KimiAudioConfig):
        super().__init__()
        self.config = config
        self.model = KimiDecoderModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.mimo_output = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    def post_init(self):
        std = self.config.initializer_range
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=std)
            elif isinstance(m, RMSNorm):
                nn.init.ones_(m.weight)

    def forward(self, text_input_ids=None, audio_input_ids=None, audio_input_embs=None, attention_mask=None,
                position_ids=None, labels=None, sentence_lens=None, num_sentence=None, shift_labels=None,
                compute_audio_logits: bool = False, ce_chunk_tokens: int = 4096, ce_compact=False,
                labelled_rows_max=None, **unused):
        """`audio_input_ids` / `text_input_ids` int64 [B, T]: the two aligned token streams of the Kimi-Audio prompt
        format (`processing_kimi_audio.py:112-116`); `audio_input_embs` [B, T, H] replaces the audio-token embeddings
        where the caller has merged continuous Whisper features into them.  `attention_mask` = document ids."""
        emb = self.model.embed_tokens
        x = audio_input_embs if audio_input_embs is not None else emb(audio_input_ids)
        if text_input_ids is not None:
            x = x + emb(text_input_ids)                                              # `:1030-1033`
        h, mimo = self.model(x, position_ids=position_ids, attention_mask=attention_mask,
                             with_mimo=compute_audio_logits)
        audio_logits = self.mimo_output(mimo) if compute_audio_logits else None
        if labelled_rows_max is not None and ce_compact is not True:
            ce_compact = (int(labelled_rows_max) + 255) // 256 * 256
        if labels is None:
            if getattr(self, "_tn_loss_parallel", None) is not None:
                raise RuntimeError("loss parallel: the head is vocabulary-sharded, use the fused lm_head + CE (pass labels)")
            return SimpleNamespace(logits=self.lm_head(h), audio_logits=audio_logits, loss=None)
        from touchnet_amd.loss.cross_entropy import fused_linear_cross_entropy
        lp = getattr(self, "_tn_loss_parallel", None)
        loss, per_token, acc = fused_linear_cross_entropy(h, self.lm_head.weight, labels, sentence_lens, num_sentence,
                                                          chunk_tokens=ce_chunk_tokens, compact=ce_compact, tp=lp)
        return SimpleNamespace(logits=None, audio_logits=audio_logits, loss=loss, loss_per_token=per_token, acc=acc)
