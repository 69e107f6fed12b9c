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
over V = 168448 on the labelled rows) AND — round 4 — the audio-input side of `prepare_audio_input_embs` (`:933-985`):
the Whisper-large-v3 speech encoder on `whisper_input_features` (`WhisperSpeechEncoder`: the Qwen2-Audio tower's blocks
with BIDIRECTIONAL attention and no pooling), the x4 reshape + `vq_adaptor` (`:322-334`), the embeddings of the discrete
speech-tokenizer ids x sqrt(2), and the scatter between the media markers (index arithmetic, no host sync).  Still the
loader's: the GLM-4-voice VQ tokenizer (`WhisperVQEncoder`, frozen, `:859-860, 957-963`) — its ids enter as
`speech_tokenizer_ids`.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from ..llama.configuration import DecoderConfig
from ..llama.modeling_llama import DecoderLayer, RMSNorm, RotaryEmbedding
from ..qwen2_audio.modeling_qwen2_audio import AudioEncoderConfig, LayerNorm, Qwen2AudioEncoder
from ..backend import ops


@dataclass
class KimiAudioConfig(DecoderConfig):
    """examples/audio/sft/asr/wenetspeech/config/Kimi-Audio-7B.json (the decoder keys)."""
    kimia_mimo_layers: int = 6
    kimia_mimo_transformer_from_layer_index: int = 21
    kimia_token_offset: int = 152064
    kimia_media_begin: int = 151661
    kimia_media_end: int = 151663
    # the audio-input side (Kimi-Audio-7B.json: use_whisper_feature, kimia_adaptor_input_dim, speech_encoder_config)
    use_whisper_feature: bool = False
    kimia_adaptor_input_dim: int = 5120
    speech_encoder_config: Optional[dict] = None      # WhisperConfig keys; None = whisper-large-v3's encoder

    def __post_init__(self):
        self.model_type = "qwen2"            # Qwen2DecoderLayer: q/k/v carry a bias
        super().__post_init__()
        self.attention_bias = True

    def speech_encoder_dims(self) -> AudioEncoderConfig:
        return AudioEncoderConfig.from_dict(self.speech_encoder_config or {})     # (defaults = whisper-large-v3)


class WhisperSpeechEncoder(Qwen2AudioEncoder):
    """transformers' WhisperEncoder as `MoonshotKimiaForCausalLM.speech_encoder` (`:337-339, 858`): the conv stem, sinusoidal
    position table, pre-LN layers and final LayerNorm of the Qwen2-Audio tower under the same parameter names — with
    Whisper's own BIDIRECTIONAL self-attention and without the tower's 2:1 pooling.  The HF encoder ignores its
    `attention_mask` (it runs on all frames of the 30 s-padded clip); so does this one."""

    def __init__(self, cfg: AudioEncoderConfig):
        super().__init__(cfg, causal=False)

    def forward(self, input_features, attention_mask=None):
        """mel [n, num_mel_bins, Tm] -> [n, (Tm - 1) // 2 + 1, d_model]"""
        h = self.stem(input_features)
        n, T, _ = h.shape
        mask = ops().causal_mask(n, T, h.device)          # one document per clip (the attention itself is not causal)
        delta, residual = None, h
        for layer in self.layers:
            delta, residual = layer(delta, residual, mask)
        return self.layer_norm(delta, residual)[0]


class VQAdaptor(nn.Module):
    """`:322-334`: Linear(4 d_enc -> H) + SiLU + Dropout(0) + Linear(H -> H) + LayerNorm(H, eps = rms_norm_eps); parameter
    names `layers.0.*`, `layers.3.*`, `layers.4.*` (the reference's nn.Sequential indices)."""

    def __init__(self, config: KimiAudioConfig):
        super().__init__()
        H = config.hidden_size
        self.layers = nn.ModuleList([nn.Linear(config.kimia_adaptor_input_dim, H), nn.Identity(), nn.Identity(),
                                     nn.Linear(H, H), LayerNorm(H, config.rms_norm_eps)])

    def forward(self, x):
        lin = lambda t, m: ops().linear_group(t, [(m.weight, m.bias)], wgrad="nt", dgrad_tn=False)[0]
        return self.layers[4](lin(torch.nn.functional.silu(lin(x, self.layers[0])), self.layers[3]))


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
        if config.use_whisper_feature:                                   # `:392-394`
            self.vq_adaptor = VQAdaptor(config)

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
        self.speech_encoder = WhisperSpeechEncoder(config.speech_encoder_dims()) if config.use_whisper_feature else None

    def prepare_audio_input_embs(self, audio_input_ids, audio_input_embs, whisper_input_features, speech_tokenizer_ids):
        """`:933-985`.  whisper_input_features [n, mel, frames]: one 30 s-padded clip per media-marker pair, in the order
        the pairs appear in the batch (the reference batch has one pair per row); speech_tokenizer_ids int64 [n, frames // 8]:
        what the frozen GLM-4-voice tokenizer returns for the same features (WITHOUT `kimia_token_offset`, added here like
        `:963`).  Positions strictly between `<|im_media_begin|>` and `<|im_media_end|>` receive
        (vq_adaptor(encoder x4-stacked) + embedding(ids)) * sqrt(2), frame k of the clip at the k-th such position."""
        enc = self.speech_encoder(whisper_input_features)                           # [n, 1500, d]
        n, Tf, d = enc.shape
        enc = enc.reshape(n, Tf // 4, d * 4)                                        # `:948-952`
        cont = self.model.vq_adaptor(enc)                                           # [n, S, H]
        ids = speech_tokenizer_ids.to(audio_input_ids.device) + self.config.kimia_token_offset
        # (`:957-963`: the tokenizer itself is frozen; its ids are data)
        speech = ((cont + self.model.embed_tokens(ids)) * (2.0 ** 0.5)).to(audio_input_embs.dtype)     # `:964-968`
        S = speech.shape[1]
        # between-markers mask + which (clip, frame) lands where: running counts over the flattened batch (no host sync)
        flat = audio_input_ids.reshape(-1)
        begin, end = flat == self.config.kimia_media_begin, flat == self.config.kimia_media_end
        nb, ne = torch.cumsum(begin, 0), torch.cumsum(end, 0)
        inside = (nb > ne) & ~begin                                                 # strictly between the two markers
        clip = (nb - 1).clamp_(0, n - 1)
        pos = torch.arange(flat.numel(), device=flat.device)
        start = torch.cummax(torch.where(begin, pos, torch.zeros_like(pos)), 0).values      # position of the open marker
        frame = (pos - start - 1).clamp_(0, S - 1)
        src = speech.reshape(n * S, -1).index_select(0, clip * S + frame)           # [B*T, H]
        out = torch.where(inside.unsqueeze(-1), src, audio_input_embs.reshape(flat.numel(), -1))
        return out.view_as(audio_input_embs)

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
            elif isinstance(m, LayerNorm):
                m.reset_parameters()
            elif isinstance(m, nn.Conv1d):
                nn.init.normal_(m.weight, mean=0.0, std=std)
                nn.init.zeros_(m.bias)

    def forward(self, text_input_ids=None, audio_input_ids=None, audio_input_embs=None, attention_mask=None,
                position_ids=None, labels=None, sentence_lens=None, num_sentence=None, shift_labels=None,
                compute_audio_logits: bool = False, ce_chunk_tokens: int = 4096, ce_compact=False,
                labelled_rows_max=None, whisper_input_features=None, whisper_attention_mask=None,
                speech_tokenizer_ids=None, **unused):
        """`audio_input_ids` / `text_input_ids` int64 [B, T]: the two aligned token streams of the Kimi-Audio prompt
        format (`processing_kimi_audio.py:112-116`); `audio_input_embs` [B, T, H] replaces the audio-token embeddings
        where the caller has merged continuous Whisper features into them.  `attention_mask` = document ids.
        `whisper_input_features` (+ `speech_tokenizer_ids`): the reference batch's own keys (`:1022-1029`): with
        `use_whisper_feature` the speech encoder / adaptor / marker scatter run here (`prepare_audio_input_embs`);
        `whisper_attention_mask` is accepted and — like in transformers' WhisperEncoder — not used."""
        emb = self.model.embed_tokens
        x = audio_input_embs if audio_input_embs is not None else emb(audio_input_ids)
        if self.speech_encoder is not None and whisper_input_features is not None:
            if speech_tokenizer_ids is None:
                raise ValueError("whisper_input_features need speech_tokenizer_ids: the frozen GLM-4-voice tokenizer "
                                 "(modeling_kimi_audio.py:957-963) runs in the loader, its ids are an input here")
            x = self.prepare_audio_input_embs(audio_input_ids, x, whisper_input_features, speech_tokenizer_ids)
        elif whisper_input_features is not None:
            raise ValueError("whisper_input_features passed to a model built with use_whisper_feature=False")
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
