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
    """Reference MFU convention (touchnet/models/llama/__init__.py:39-54): dense-parameter term 6*N plus the
    attention-score term 12 * layers * heads * head_dim * T — 2 matmuls forward + 4 backward, x2 for
    multiply-add, NO discount for causal / packing sparsity, NO credit for recomputation."""
    text = getattr(model_config, "text_config", model_config)
    head_dim = text.hidden_size // text.num_attention_heads
    attention_term = 12 * text.num_hidden_layers * text.num_attention_heads * head_dim * seq_len
    return 6 * num_params + attention_term


def get_num_params(model: torch.nn.Module, exclude_embedding: bool = False) -> int:
    """Parameter count as touchnet/models/llama/__init__.py:57-67 defines it (tied weights once; with
    `exclude_embedding` the nn.Embedding children of the base model are left out — that is N_wo_emb of the MFU)."""
    lm = getattr(model, "language_model", model)
    unique = {id(p): p.numel() for p in model.parameters()}
    total = sum(unique.values())
    if not exclude_embedding:
        return total
    base = getattr(lm, getattr(lm, "base_model_prefix", "model"))
    emb = [m for m in base.children() if isinstance(m, torch.nn.Embedding)]
    return total - sum(p.numel() for m in emb for p in m.parameters())
