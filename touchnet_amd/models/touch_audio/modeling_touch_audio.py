"""LlamaForASR / TouchAudioForCausalLM on the HIP path.

Mirrors touchnet/models/touch_audio/modeling_touch_audio.py:19-152 and
configuration_touch_audio.py:8-58: a bias-free `projector` Linear(input_size -> hidden) whose output is
ADDED to the token embeddings (`:123-131`; zero feature rows stay zero, docs/TouchAudioForCausalLM.md:10),
then the packed causal LM.  Parameter names: `projector.weight`, `language_model.model.*`,
`language_model.lm_head.weight`.  Differences, all host-side: the projector GEMM and the add are one
addmm; the NaN guard of `:133-134` (a host sync per step) is opt-in via `check_nan`.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import torch
import torch.nn as nn

from ..llama.configuration import DecoderConfig
from ..llama.modeling_llama import PackedCausalLM


@dataclass
class TouchAudioConfig:
    text_config: DecoderConfig = field(default_factory=DecoderConfig)
    input_size: int = 4096                     # audio_config.input_size = num_mel_bins * stack_length
    pad_token_id: int = None

    @classmethod
    def from_dict(cls, d):
        tc = d.get("text_config", {})
        tc = tc if isinstance(tc, DecoderConfig) else DecoderConfig.from_dict(tc)
        return cls(text_config=tc, input_size=d.get("audio_config", {}).get("input_size", 4096),
                   pad_token_id=d.get("pad_token_id"))

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))


class TouchAudioForCausalLM(nn.Module):
    config_class = TouchAudioConfig
    base_model_prefix = "language_model"

    def __init__(self, config: TouchAudioConfig, check_nan: bool = False):
        super().__init__()
        self.config = config
        self.projector = nn.Linear(config.input_size, config.text_config.hidden_size, bias=False)
        self.language_model = PackedCausalLM(config.text_config)
        self.check_nan = check_nan

    def post_init(self):
        self.language_model.post_init()
        nn.init.normal_(self.projector.weight, mean=0.0, std=self.config.text_config.initializer_range)

    def forward(self, input_ids=None, input_features=None, attention_mask=None, position_ids=None,
                inputs_embeds=None, **loss_kwargs):   # loss_kwargs: labels / sentence_lens / num_sentence / context_parallel
        loss_kwargs.pop("shift_labels", None)                                   # (a key of the reference's batches)
        if inputs_embeds is None and input_ids is None and input_features is not None:
            # audio pretraining from the UNPACKED batcher (`batch_audio`, processing_touch_audio.py:260): no token ids at
            # all — the frames alone are the input (the reference's forward has no branch for it)
            B, T, _ = input_features.shape
            w = self.projector.weight
            inputs_embeds = (input_features.to(w.dtype).reshape(B * T, -1) @ w.t()).view(B, T, -1)
        if inputs_embeds is None:
            emb = self.language_model.model.embed_tokens(input_ids)             # [B, T, H]
            B, T, H = emb.shape
            if input_features is None:                                          # `:128-130`: keep the projector in the graph
                input_features = torch.zeros(B, T, self.config.input_size, device=emb.device, dtype=emb.dtype)
            feats = input_features.to(emb.dtype).reshape(B * T, -1)
            inputs_embeds = torch.addmm(emb.reshape(B * T, H), feats, self.projector.weight.t()).view(B, T, H)
        if self.check_nan and torch.isnan(inputs_embeds).any():
            raise ValueError("NaN in data.")
        return self.language_model(inputs_embeds=inputs_embeds, position_ids=position_ids,
                                   attention_mask=attention_mask, **loss_kwargs)
