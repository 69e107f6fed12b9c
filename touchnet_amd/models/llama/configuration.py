"""Decoder configuration read from the SAME Hugging Face JSON files the reference recipes use
(examples/**/config/*.json, tests/assets/config/tiny_llama.json): no transformers dependency."""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class DecoderConfig:
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = None
    head_dim: Optional[int] = None
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None
    tie_word_embeddings: bool = False
    attention_bias: bool = False          # Qwen2: q/k/v carry a bias (o_proj does not)
    initializer_range: float = 0.02
    model_type: str = "llama"
    pad_token_id: Optional[int] = None
    bos_token_id: Optional[int] = None
    eos_token_id: Optional[int] = None
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        if self.model_type == "qwen2":
            self.attention_bias = True

    @classmethod
    def from_dict(cls, d: dict) -> "DecoderConfig":
        known = {k: d[k] for k in cls.__dataclass_fields__ if k in d and k != "extra"}
        cfg = cls(**known)
        cfg.extra = {k: v for k, v in d.items() if k not in known}
        return cfg

    @classmethod
    def from_json_file(cls, path: str) -> "DecoderConfig":      # same entry point as train.py:127
        with open(path) as f:
            return cls.from_dict(json.load(f))

    def to_dict(self) -> dict:
        d = {k: getattr(self, k) for k in self.__dataclass_fields__ if k != "extra"}
        d.update(self.extra)
        return d
