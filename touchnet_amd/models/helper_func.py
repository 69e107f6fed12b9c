"""FSDP2 wrapping of the packed models (mirrors touchnet/models/helper_func.py:134-202 and the per-model
grouping of parallelize_llama.py:64-80 / parallelize_touch_audio.py:72-98 / parallelize_qwen2_audio.py:51-80).

Shard groups: every decoder block (and every audio-tower block) is one `fully_shard` unit, the remaining
parameters (embeddings, lm_head, final norm, conv stem, projector) form the root unit.  Mixed precision =
bf16 parameters for compute, fp32 gradient reduction, as the reference configures it.  MI355X-specific
choice: 288 GB of HBM per GPU holds the whole bf16 model next to the optimizer shards, so the default
policy keeps parameters unsharded between forward and backward (`reshard_after_forward=False`, ZeRO-2
traffic: one all-gather + one reduce-scatter per block per step instead of two all-gathers); pass
"default" to get the reference's ZeRO-3 schedule.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _fully_shard():
    try:
        from torch.distributed.fsdp import MixedPrecisionPolicy, fully_shard
    except ImportError:                                   # torch < 2.6 location
        from torch.distributed._composable.fsdp import MixedPrecisionPolicy, fully_shard
    return fully_shard, MixedPrecisionPolicy


def block_groups(model: nn.Module):
    """Lists of transformer blocks, in the order the reference wraps them."""
    groups = []
    lm = getattr(model, "language_model", model)
    groups.append(list(lm.model.layers))
    if hasattr(lm.model, "mimo_layers"):                   # Kimi-Audio's second stack (parallelize_kimi_audio.py)
        groups.append(list(lm.model.mimo_layers))
    if hasattr(model, "audio_tower"):
        groups.append(list(model.audio_tower.layers))
    return groups


def apply_fsdp(model: nn.Module, dp_mesh, param_dtype=torch.bfloat16, reduce_dtype=torch.float32,
               pp_enabled: bool = False, cpu_offload: bool = False, reshard_after_forward_policy: str = "never"):
    """Positional order of the reference (helper_func.py:134-145).  `pp_enabled` / `cpu_offload` must be False here."""
    if pp_enabled or cpu_offload:
        raise NotImplementedError("pipeline parallelism / CPU offload are outside the MI355X path")
    fully_shard, MixedPrecisionPolicy = _fully_shard()
    if any(p.is_cuda for p in model.parameters()):
        # FSDP2 copies a returned weight gradient into its reduce-scatter input behind the backward's own stream: a product
        # computed on the optional weight-gradient side stream must be waited for there (functional._beside)
        from touchnet_amd import functional as F
        F.WGRAD_RETURNS_NEED_SYNC = True
    cfg = {"mesh": dp_mesh, "mp_policy": MixedPrecisionPolicy(param_dtype=param_dtype, reduce_dtype=reduce_dtype)}
    for blocks in block_groups(model):
        for i, blk in enumerate(blocks):
            if reshard_after_forward_policy == "always":
                reshard = True
            elif reshard_after_forward_policy == "never":
                reshard = False
            elif reshard_after_forward_policy == "default":
                reshard = i < len(blocks) - 1             # helper_func.py:186-191
            else:
                raise ValueError(f"Invalid reshard_after_forward_policy: {reshard_after_forward_policy}.")
            fully_shard(blk, **cfg, reshard_after_forward=reshard)
    fully_shard(model, **cfg, reshard_after_forward=reshard_after_forward_policy != "never")
    return model
