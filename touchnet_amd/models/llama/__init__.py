"""Llama adapter: mirrors touchnet/models/llama/__init__.py (pre_init / post_init / flop + param counts)."""
import torch

from .configuration import DecoderConfig
from .modeling_llama import PackedCausalLM, RotaryEmbedding


def pre_init(args=None):
    """touchnet/models/llama/__init__.py:11-15 patches HF classes with liger here.  Our model class IS the
    MI355X path, so the only job left is to make sure the HIP library is loadable before any allocation."""
    from touchnet_amd import _C
    _C.lib()


def post_init(model: PackedCausalLM, init_device: torch.device):
    """touchnet/models/llama/__init__.py:19-36: re-derive rope inv_freq (buffers are not materialised by
    `to_empty`), reset norm weights to 1, NaN/Inf check on the parameters."""
    lm = getattr(model, "language_model", model)
    rot = lm.model.rotary_emb
    rot.inv_freq = RotaryEmbedding.compute_inv_freq(rot.config, device=init_device)
    torch.nn.init.ones_(lm.model.norm.weight)
    for layer in lm.model.layers:
        torch.nn.init.ones_(layer.input_layernorm.weight)
        torch.nn.init.ones_(layer.post_attention_layernorm.weight)
    for name, p in model.named_parameters():
        if not torch.isfinite(p).all():
            raise ValueError(f"NaN/inf in model parameters `{name}`.")


def get_num_flop_per_token(num_params: int, model_config, seq_len: int) -> int:
    """6*N + 12*L*H*Dh*T, the reference's MFU convention (touchnet/models/llama/__init__.py:39-54):
    no causal / packing sparsity discount, no recompute credit."""
    cfg = getattr(model_config, "text_config", model_config)
    l, h = cfg.num_hidden_layers, cfg.num_attention_heads
    q = cfg.hidden_size // cfg.num_attention_heads
    return 6 * num_params + 12 * l * h * q * seq_len


def get_num_params(model: torch.nn.Module, exclude_embedding: bool = False) -> int:
    """touchnet/models/llama/__init__.py:57-67 (embedding = nn.Embedding children of the base model)."""
    lm = getattr(model, "language_model", model)
    seen, total = set(), 0
    for p in model.parameters():
        if id(p) not in seen:
            seen.add(id(p))
            total += p.numel()
    if exclude_embedding:
        sub = getattr(lm, getattr(lm, "base_model_prefix", "model"))
        total -= sum(sum(p.numel() for p in m.parameters()) for m in sub.children()
                     if isinstance(m, torch.nn.Embedding))
    return total
