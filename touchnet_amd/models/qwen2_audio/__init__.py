"""Qwen2-Audio adapter (mirrors touchnet/models/qwen2_audio/__init__.py:252-320)."""
import torch

from ..llama import RotaryEmbedding
from .modeling_qwen2_audio import (AudioEncoderConfig, Qwen2AudioConfig,  # noqa: F401
                                   Qwen2AudioPackedForConditionalGeneration)


def pre_init(args=None):
    from touchnet_amd import _C
    _C.lib()


def post_init(model, init_device: torch.device):
    """qwen2_audio/__init__.py:263-288: reset LayerNorms of the tower, rope table + RMSNorm weights of the LM."""
    model.audio_tower.layer_norm.reset_parameters()
    for layer in model.audio_tower.layers:
        layer.self_attn_layer_norm.reset_parameters()
        layer.final_layer_norm.reset_parameters()
    lm = model.language_model
    lm.model.rotary_emb.inv_freq = RotaryEmbedding.compute_inv_freq(lm.model.rotary_emb.config, device=init_device)
    torch.nn.init.ones_(lm.model.norm.weight)
    for layer in lm.model.layers:
        torch.nn.init.ones_(layer.input_layernorm.weight)
        torch.nn.init.ones_(layer.post_attention_layernorm.weight)
    for name, p in model.named_parameters():
        if not torch.isfinite(p).all():
            raise ValueError(f"NaN/inf in model parameters `{name}`.")


def get_num_flop_per_token(num_params: int, model_config, seq_len: int) -> int:
    """qwen2_audio/__init__.py:291-307: 6*N + 12*L*H*Dh*T over the TEXT tower (audio tower only through N)."""
    c = model_config.text_config
    return 6 * num_params + 12 * c.num_hidden_layers * c.num_attention_heads * (
        c.hidden_size // c.num_attention_heads) * seq_len


def get_num_params(model: torch.nn.Module, exclude_embedding: bool = False) -> int:
    """qwen2_audio/__init__.py:310-320 (only the language model's nn.Embedding is excluded)."""
    total = sum(p.numel() for p in model.parameters())
    if exclude_embedding:
        sub = model.language_model.model
        total -= sum(sum(p.numel() for p in m.parameters()) for m in sub.children()
                     if isinstance(m, torch.nn.Embedding))
    return total
