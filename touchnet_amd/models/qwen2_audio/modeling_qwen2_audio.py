"""Qwen2-Audio on the HIP path, PACKED: Whisper-style audio tower -> projector -> audio features written
into the packed token sequence -> Qwen2 decoder with document-masked attention.

What the reference runs (padded, un-packed):
  tower    touchnet/models/qwen2_audio/__init__.py:18-133 (conv stem, tiled positions :52-73, 32 pre-LN
           layers with attention FORCED causal :190-193, avg-pool(2), LayerNorm)
  merge    :186-229 (projector, boolean compaction of valid frames, masked_scatter at the AUDIO tokens)
  decoder  :231-249 (plain causal SDPA per padded row)
Here several samples share one packed row; the decoder sees their document ids, which gives every sample
exactly the causal attention it had in its own padded row (the equivalence
tests/touchnet/utils/test_pack_loss.py proves for the loss).  Parameter names follow
Qwen2AudioForConditionalGeneration (transformers 4.51.3): `audio_tower.*`,
`multi_modal_projector.linear.*`, `language_model.*`.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as TF

from ..backend import ops
from ..llama.configuration import DecoderConfig
from ..llama.modeling_llama import PackedCausalLM


@dataclass
class AudioEncoderConfig:
    num_mel_bins: int = 128
    d_model: int = 1280
    encoder_layers: int = 32
    encoder_attention_heads: int = 20
    encoder_ffn_dim: int = 5120
    max_source_positions: int = 1500
    init_std: float = 0.02

    @classmethod
    def from_dict(cls, d):
        return cls(**{k: d[k] for k in cls.__dataclass_fields__ if k in d})


@dataclass
class Qwen2AudioConfig:
    audio_config: AudioEncoderConfig = field(default_factory=AudioEncoderConfig)
    text_config: DecoderConfig = field(default_factory=lambda: DecoderConfig(model_type="qwen2"))
    audio_token_index: int = 151646

    @classmethod
    def from_dict(cls, d):
        tc = dict(d.get("text_config", {}))
        tc.setdefault("model_type", "qwen2")
        return cls(audio_config=AudioEncoderConfig.from_dict(d.get("audio_config", {})),
                   text_config=DecoderConfig.from_dict(tc), audio_token_index=d.get("audio_token_index", 151646))

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim)), eps

    def forward(self, x, residual=None):
        return ops().layer_norm(x, self.weight, self.bias, self.eps, residual=residual)

    def reset_parameters(self):
        nn.init.ones_(self.weight)
        nn.init.zeros_(self.bias)


class EncoderAttention(nn.Module):
    def __init__(self, dim, heads, causal: bool = True):
        super().__init__()
        self.num_heads, self.head_dim, self.causal = heads, dim // heads, causal
        self.k_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=True)
        self.q_proj = nn.Linear(dim, dim, bias=True)
        self.out_proj = nn.Linear(dim, dim, bias=True)

    def forward(self, x, mask):
        B, T, C = x.shape
        # one autograd node for q/k/v: at 1280 x 1280 a separate weight-gradient GEMM is 25 output tiles (351 TFLOP/s),
        # the three fused run at 781 (functional._LinearGroup, wgrad="nt_fused"); parameters keep the HF names
        q, k, v = ops().linear_group(x, [(self.q_proj.weight, self.q_proj.bias), (self.k_proj.weight, None),
                                         (self.v_proj.weight, self.v_proj.bias)], wgrad="nt_fused", dgrad_tn=False)
        q = q.view(B, T, self.num_heads, self.head_dim)
        k = k.view(B, T, self.num_heads, self.head_dim)
        v = v.view(B, T, self.num_heads, self.head_dim)
        # (causal: the Qwen2-Audio tower as the reference forces it; bidirectional: Whisper's own encoder, Kimi-Audio)
        attend = ops().packed_attention if self.causal else ops().bidirectional_attention
        a = attend(q, k, v, mask, self.head_dim ** -0.5)
        return ops().linear_group(a.view(B, T, C), [(self.out_proj.weight, self.out_proj.bias)], wgrad="nt",
                                  dgrad_tn=False)[0]


class EncoderLayer(nn.Module):
    def __init__(self, cfg: AudioEncoderConfig, causal: bool = True):
        super().__init__()
        self.self_attn = EncoderAttention(cfg.d_model, cfg.encoder_attention_heads, causal)
        self.self_attn_layer_norm = LayerNorm(cfg.d_model)
        self.fc1 = nn.Linear(cfg.d_model, cfg.encoder_ffn_dim)
        self.fc2 = nn.Linear(cfg.encoder_ffn_dim, cfg.d_model)
        self.final_layer_norm = LayerNorm(cfg.d_model)

    def forward(self, delta, residual, mask):
        if delta is None:
            x = self.self_attn_layer_norm(residual)
        else:
            x, residual = self.self_attn_layer_norm(delta, residual)
        a = self.self_attn(x, mask)
        x, residual = self.final_layer_norm(a, residual)
        # fc2(gelu(fc1(x))) as one autograd node: GELU in fc1's epilogue, its backward in the epilogue of fc2's input-gradient
        # product (functional._GeluMLP)
        return ops().gelu_mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias), residual


# TN_TOWER_VALID_FRAMES=0 restores the reference's schedule (all 1500 frames of every padded clip through the tower).
TOWER_VALID_FRAMES_ONLY = os.environ.get("TN_TOWER_VALID_FRAMES", "1") != "0"
# TN_TOWER_CONV=miopen: the conv stem through torch's conv1d (MIOpen) instead of the hand-written GEMM (A/B switch)
TOWER_CONV_GEMM = os.environ.get("TN_TOWER_CONV", "own") != "miopen"


class Qwen2AudioEncoder(nn.Module):
    def __init__(self, cfg: AudioEncoderConfig, causal: bool = True):
        super().__init__()
        self.config = cfg
        self.conv1 = nn.Conv1d(cfg.num_mel_bins, cfg.d_model, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(cfg.d_model, cfg.d_model, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(cfg.max_source_positions, cfg.d_model)
        self.embed_positions.requires_grad_(False)
        self.layers = nn.ModuleList([EncoderLayer(cfg, causal) for _ in range(cfg.encoder_layers)])
        self.layer_norm = LayerNorm(cfg.d_model)

    def positions(self, seq_len):
        """qwen2_audio/__init__.py:52-73: slice the table, or tile it for audio longer than 30 s."""
        pos = self.embed_positions.weight
        n = pos.shape[0]
        if n >= seq_len:
            return pos[:seq_len]
        reps, rem = divmod(seq_len, n)
        return torch.cat([pos] * reps + ([pos[:rem]] if rem else []), dim=0)

    def stem(self, input_features):
        """mel [n, num_mel_bins, Tm] -> conv stem + positions [n, (Tm - 1) // 2 + 1, d_model]"""
        x = input_features.to(self.conv1.weight.dtype)
        if TOWER_CONV_GEMM and x.is_cuda and x.dtype == torch.bfloat16 and self.conv1.weight.shape[0] % 64 == 0:
            # both convolutions as GEMMs of the hand-written kernel over channels-last im2col VIEWS (functional._Conv1dK3);
            # a loader that keeps the mel as [n, Tm, bins] (the frontend's own layout) pays no transpose here
            h, t1 = ops().conv1d_k3(x.transpose(1, 2), self.conv1.weight, self.conv1.bias, 1, need_dx=False)
            h, t2 = ops().conv1d_k3(ops().gelu(h)[:, :t1], self.conv2.weight, self.conv2.bias, 2)
            h = ops().gelu(h)[:, :t2]
            return h + self.positions(t2)[None].to(h.dtype)
        x = ops().gelu(self.conv1(x))
        x = ops().gelu(self.conv2(x))
        h = x.permute(0, 2, 1).contiguous()
        return h + self.positions(h.shape[1])[None].to(h.dtype)

    def encode(self, h, mask):
        """the encoder layers, 2:1 average pooling and the final LayerNorm on frames [n, T, d] (T even per document)"""
        delta, residual = None, h
        for layer in self.layers:
            delta, residual = layer(delta, residual, mask)
        h = residual + delta
        h = TF.avg_pool1d(h.permute(0, 2, 1), 2, 2).permute(0, 2, 1).contiguous()
        return self.layer_norm(h)

    def forward(self, input_features):
        """mel [n, num_mel_bins, Tm] -> [n, Tm // 4, d_model]"""
        h = self.stem(input_features)
        n, T, _ = h.shape
        return self.encode(h, ops().causal_mask(n, T, h.device))   # is_causal forced True (:190-193), per sample

    def forward_valid(self, input_features, out_lengths, total: int):
        """Only the frames that reach the language model: -> [total, d_model], rows [0, len_i) of clip i, clip after clip
        (= what `:202-205`'s boolean compaction keeps of forward()'s output).

        The reference runs the 32 layers on all 1500 frames of every 30 s-PADDED clip and throws the padded part away
        afterwards (`:202-205`).  Because it also forces the tower's attention to be causal (`:190-193`) and everything
        else in a layer acts per frame, the frames a clip keeps — [0, 2 len_i): output token j pools frames 2j, 2j + 1 —
        do not depend on the frames behind them.  So the conv stem (3 ms) still sees the padded clips, but the layers run
        on the kept frames only, packed clip after clip into one row with one document id per clip — the same
        document-masked attention the decoder uses.  Same outputs (tested against the padded oracle), and for
        WenetSpeech-length utterances (2-14.5 s) 3.6x fewer tower frames.  `total` = sum(out_lengths) is known from the
        number of AUDIO positions: no data-dependent shape, no host sync."""
        h = self.stem(input_features)
        n, T, C = h.shape
        fl = 2 * out_lengths.to(torch.int64)
        ends = torch.cumsum(fl, 0)
        idx = torch.arange(2 * total, device=h.device)
        clip = torch.searchsorted(ends, idx, right=True).clamp_(max=n - 1)
        src = (clip * T + (idx - (ends - fl)[clip])).clamp_(0, n * T - 1)
        hp = h.reshape(n * T, C).index_select(0, src)[None]                    # [1, 2 total, C]
        mask = ops().build_packed_mask((clip + 1)[None])
        return self.encode(hp, mask)[0]


class MultiModalProjector(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.linear = nn.Linear(d_in, d_out, bias=True)

    def forward(self, x):
        return self.linear(x)


class Qwen2AudioPackedForConditionalGeneration(nn.Module):
    config_class = Qwen2AudioConfig
    base_model_prefix = "language_model"

    def __init__(self, config: Qwen2AudioConfig):
        super().__init__()
        self.config = config
        self.audio_tower = Qwen2AudioEncoder(config.audio_config)
        self.multi_modal_projector = MultiModalProjector(config.audio_config.d_model, config.text_config.hidden_size)
        self.language_model = PackedCausalLM(config.text_config)

    def post_init(self):
        self.language_model.post_init()
        std = self.config.audio_config.init_std
        for m in list(self.audio_tower.modules()) + list(self.multi_modal_projector.modules()):
            if isinstance(m, (nn.Linear, nn.Conv1d)):
                nn.init.normal_(m.weight, mean=0.0, std=std)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=std)
            elif isinstance(m, LayerNorm):
                m.reset_parameters()

    def forward(self, input_ids=None, input_features=None, audio_output_lengths=None, audio_positions=None,
                attention_mask=None, position_ids=None, audio_rows=None, feature_attention_mask=None, **loss_kwargs):
        """input_ids [B, T] packed, AUDIO placeholder tokens where audio features go;
        input_features [n_audio, n_mels, Tm]; audio_output_lengths int64 [n_audio] (valid tokens per audio,
        `((L-1)//2+1-2)//2+1`, processing_qwen2_audio.py:79-82); audio_positions int64 [sum(lengths)] flat
        indices into B*T (computed from input_ids when omitted — that costs a host sync).
        `audio_rows` int64 [n_positions] (context parallelism, utils.context_parallel.ContextParallel.shard_audio):
        `input_ids` is this rank's part of the sequence, `input_features` the clips that touch it, and row
        `audio_rows[j]` of the tower's [n * Ta] output rows goes to position `audio_positions[j]`."""
        emb = self.language_model.model.embed_tokens(input_ids)
        B, T, H = emb.shape
        loss_kwargs.pop("shift_labels", None)                # (a key of the reference batch for its liger path)
        if feature_attention_mask is not None and audio_output_lengths is None:
            # the reference's own (unpacked) batch, touchnet/models/qwen2_audio/processing_qwen2_audio.py:119-147: valid mel
            # frames per clip -> audio tokens per clip (`_get_feat_extract_output_lengths`, __init__.py:184-186); its 0/1
            # `attention_mask` already reads as document ids here (1 = the row's one document, 0 = padding)
            frames = feature_attention_mask.sum(-1)
            audio_output_lengths = (((frames - 1) // 2 + 1) - 2) // 2 + 1
        if input_features is not None:
            if audio_positions is None:                      # (costs a host sync: loaders should supply the positions)
                audio_positions = (input_ids.reshape(-1) == self.config.audio_token_index).nonzero().squeeze(1)
            n = input_features.shape[0]
            Ta = ((input_features.shape[-1] - 1) // 2 + 1) // 2
            packed_tower = (TOWER_VALID_FRAMES_ONLY and audio_output_lengths is not None and audio_rows is None
                            and 0 < audio_positions.numel() < n * Ta)
            if packed_tower:                                 # clips shorter than their padding: skip the padded frames
                feats = self.multi_modal_projector(self.audio_tower.forward_valid(
                    input_features, audio_output_lengths, audio_positions.numel()))     # [total, H], already compact
            else:
                feats = self.multi_modal_projector(self.audio_tower(input_features))    # [n, Ta, H]
                feats = feats.reshape(n * Ta, H)
            if audio_rows is not None:
                feats = feats.index_select(0, audio_rows)
            elif audio_output_lengths is not None and not packed_tower:
                # rows [0, len_i) of clip i, in clip order (`:202-205`'s boolean compaction) WITHOUT a data-dependent
                # shape: the number of valid rows is the number of AUDIO positions, known from the tensor's size
                total = audio_positions.numel()
                ends = torch.cumsum(audio_output_lengths.to(torch.int64), 0)
                if os.environ.get("TN_DEBUG_CHECKS") == "1" and int(ends[-1]) > total:
                    # the reference raises here (`__init__.py:215-219` only pads when features are FEWER); the product
                    # path cannot afford the host read-back per step, so the check is a debugging switch
                    raise ValueError(f"audio features ({int(ends[-1])}) outnumber the AUDIO tokens ({total})")
                idx = torch.arange(total, device=feats.device)
                clip = torch.searchsorted(ends, idx, right=True).clamp_(max=n - 1)
                src = clip * Ta + (idx - (ends - audio_output_lengths)[clip])
                # more AUDIO tokens than valid feature rows: the reference repeats the LAST valid row (__init__.py:215-219)
                last = (n - 1) * Ta + audio_output_lengths[-1].to(torch.int64) - 1
                src = torch.where(idx < ends[-1], src, last)
                feats = feats.index_select(0, src.clamp_(0, n * Ta - 1))
            if feats.shape[0] != audio_positions.numel():
                raise ValueError(f"audio features ({feats.shape[0]}) and audio tokens "
                                 f"({audio_positions.numel()}) mismatch")
            emb = emb.reshape(B * T, H).index_copy(0, audio_positions, feats.to(emb.dtype)).view(B, T, H)
        return self.language_model(inputs_embeds=emb, position_ids=position_ids, attention_mask=attention_mask,
                                   **loss_kwargs)
