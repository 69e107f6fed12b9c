"""Kimi-Audio adapter (mirrors touchnet/models/kimi_audio/__init__.py:25-93: pre/post init, flop + parameter counts)."""
import torch

from ..llama import RotaryEmbedding
from .modeling_kimi_audio import KimiAudioConfig, KimiAudioPackedForCausalLM, KimiDecoderModel  # noqa: F401


def pre_init(args=None):
    from touchnet_amd import _C
    _C.lib()


def post_init(model, init_device: torch.device):
    rot = model.model.rotary_emb
    rot.inv_freq = RotaryEmbedding.compute_inv_freq(rot.config, device=init_device)
    torch.nn.init.ones_(model.model.norm.weight)
    torch.nn.init.ones_(model.model.mimo_norm.weight)
    for layer in list(model.model.layers) + list(model.model.mimo_layers):
        torch.nn.init.ones_(layer.input_layernorm.weight)
        torch.nn.init.ones_(layer.post_attention_layernorm.weight)
    for name, p in model.named_parameters():
        if not torch.isfinite(p).all():
            raise ValueError(f"NaN/inf in model parameters `{name}`.")


def get_num_flop_per_token(num_params: int, model_config, seq_len: int, with_mimo: bool = True) -> int:
    """kimi_audio/__init__.py:63-80: 6*N + 12*(L + L_mimo)*H*Dh*T (the speech encoder is not counted).
    `with_mimo=False`: the same formula over what a TEXT-head training step executes — the 28 decoder layers, the final
    norm and lm_head; the mimo layers, the mimo norm and the audio head run only with `compute_audio_logits` and are left
    out (`num_params` is ignored, the executed parameters follow from the config)."""
    c = model_config
    head_dim = c.hidden_size // c.num_attention_heads
    if with_mimo:
        return 6 * num_params + 12 * (c.num_hidden_layers + c.kimia_mimo_layers) * c.num_attention_heads * head_dim * seq_len
    H, I, q, kv = c.hidden_size, c.intermediate_size, c.num_attention_heads * head_dim, c.num_key_value_heads * head_dim
    layer = H * q + q + 2 * (H * kv + kv) + q * H + 3 * H * I + 2 * H          # Qwen2 layer: biased q/k/v, two norms
    executed = c.num_hidden_layers * layer + H + c.vocab_size * H              # + final norm + lm_head
    return 6 * executed + 12 * c.num_hidden_layers * c.num_attention_heads * head_dim * seq_len


def get_num_params(model: torch.nn.Module, exclude_embedding: bool = False) -> int:
    """kimi_audio/__init__.py:83-93."""
    total = sum(p.numel() for p in model.parameters())
    if exclude_embedding:
        total -= sum(sum(p.numel() for p in m.parameters()) for m in model.model.children()
                     if isinstance(m, torch.nn.Embedding))
    return total
