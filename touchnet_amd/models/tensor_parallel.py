"""Tensor parallelism for the packed decoder blocks — BASELINE config E ("Kimi-Audio-7B … TP=2 × FSDP2 dp=4").

The reference's template is touchnet/models/llama/parallelize_llama.py:105-196 (`parallelize_module` with
Colwise / Rowwise / SequenceParallel plans over DTensors); it does not support Kimi-Audio at all.  Our
blocks do not call `nn.Linear.forward` (they hand weight tensors to multi-op HIP nodes), so the plan is applied
Megatron-style on the modules themselves, which is also what the kernels want (plain local tensors, no DTensor
dispatch on the hot path):

  attention   q / k / v projections COLUMN-parallel by heads (rank r owns query heads [r Nh/tp, (r+1) Nh/tp) and the
              kv heads they read — GQA groups never straddle ranks), o_proj ROW-parallel over the same heads
  MLP         gate / up column-parallel, down row-parallel
  per block   ONE all-reduce of the attention output and ONE of the MLP output in forward (xGMI, RCCL), their
              mirrors (all-reduce of the input gradients) in backward; the residual stream, norms, embeddings and the
              heads stay replicated over the tp ranks (no sequence parallelism in this first plan)

FSDP2 then shards the (already tp-local) parameters over the `dp_shard_cp` mesh as usual.  The optimizer has to know
which parameters are tp-sharded: their squared gradient norm is summed over the tp group, the replicated ones'
is not (`apply_tp` records the names on the model, `FusedAdamW(tp_group=, tp_param_ids=)` uses them).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn

from touchnet_amd.models.helper_func import block_groups


class EmulatedTPMesh:
    """`bench.py --tp N --emulate-rank r`: ONE process plays rank r of an N-way tensor-parallel group — it holds rank r's
    shards and runs rank r's kernels; the all-reduces (a device-to-device sum of equally shaped tensors) are skipped."""
    emulated = True

    def __init__(self, size: int, rank: int):
        self._size, self._rank = int(size), int(rank)

    def get_group(self):
        return self

    def size(self):
        return self._size

    def get_local_rank(self):
        return self._rank


def tp_all_reduce(t: torch.Tensor, group) -> None:
    if getattr(group, "emulated", False):
        return
    dist.all_reduce(t, group=group)


class _CopyToTP(torch.autograd.Function):
    """identity forward, all-reduce of the gradient backward (input of a column-parallel region)"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        tp_all_reduce(g, ctx.group)
        return g, None


class _ReduceFromTP(torch.autograd.Function):
    """all-reduce forward, identity backward (output of a row-parallel region)"""

    @staticmethod
    def forward(ctx, x, group):
        x = x.contiguous()
        tp_all_reduce(x, group)
        return x

    @staticmethod
    def backward(ctx, g):
        return g, None


def _shard(param: nn.Parameter, dim: int, rank: int, tp: int) -> nn.Parameter:
    n = param.shape[dim]
    if n % tp:
        raise ValueError(f"dimension {dim} of a {tuple(param.shape)} parameter is not divisible by tp={tp}")
    local = param.detach().narrow(dim, rank * (n // tp), n // tp)
    local = torch.empty_like(local, device="meta") if param.is_meta else local.clone()
    return nn.Parameter(local, requires_grad=param.requires_grad)


def _wrap(module: nn.Module, group) -> None:
    inner = module.forward

    def forward(x, *args, **kwargs):
        return _ReduceFromTP.apply(inner(_CopyToTP.apply(x, group), *args, **kwargs), group)
    module.forward = forward


def apply_tp(model: nn.Module, tp_mesh, loss_parallel: bool = False) -> nn.Module:
    if loss_parallel:
        raise NotImplementedError("loss parallel (vocabulary-sharded lm_head + CE) is not part of this TP plan")
    group, tp, rank = tp_mesh.get_group(), tp_mesh.size(), tp_mesh.get_local_rank()
    sharded = []
    for blocks in block_groups(model):
        for blk in blocks:
            attn, mlp = getattr(blk, "self_attn", None), getattr(blk, "mlp", None)
            if attn is None or mlp is None or not hasattr(attn, "num_kv_heads"):
                continue                                            # (the audio tower's blocks stay replicated)
            if attn.num_heads % tp or attn.num_kv_heads % tp:
                raise ValueError(f"{attn.num_heads} query / {attn.num_kv_heads} kv heads cannot be split {tp} ways")
            for lin, dim in ((attn.q_proj, 0), (attn.k_proj, 0), (attn.v_proj, 0), (attn.o_proj, 1),
                             (mlp.gate_proj, 0), (mlp.up_proj, 0), (mlp.down_proj, 1)):
                if lin.bias is not None and dim == 1:
                    # a row-parallel layer's bias would be added on every rank in front of the all-reduce (= tp times)
                    raise NotImplementedError("bias on a row-parallel layer (o_proj / down_proj) is not part of this TP plan")
                lin.weight = _shard(lin.weight, dim, rank, tp)
                sharded.append(lin.weight)
                if lin.bias is not None and dim == 0:
                    lin.bias = _shard(lin.bias, 0, rank, tp)
                    sharded.append(lin.bias)
            attn.num_heads //= tp
            attn.num_kv_heads //= tp
            _wrap(attn, group)
            _wrap(mlp, group)
    ids = {id(p) for p in sharded}
    model._tn_tp = {"group": group, "size": tp, "rank": rank,
                    "sharded_names": {n for n, p in model.named_parameters() if id(p) in ids}}
    return model


@torch.no_grad()
def reinit_tp_shards(model: nn.Module, seed: int, std: float) -> None:
    """`post_init` draws every parameter from the process RNG, which is seeded identically on all ranks: right for the
    replicated parameters, but the tp ranks' SHARDS of one weight would come out identical (duplicated heads / MLP
    columns).  Redraw the sharded ones from a generator keyed by (seed, parameter name, tp rank)."""
    info = getattr(model, "_tn_tp", None)
    if not info or info["size"] <= 1:
        return
    import zlib
    for name, p in model.named_parameters():
        if name in info["sharded_names"] and not p.is_meta:
            local = p._local_tensor if hasattr(p, "_local_tensor") else p
            g = torch.Generator(device=local.device)
            g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode()) * 31 + info["rank"]) % (2 ** 63 - 1))
            if name.endswith("bias"):
                local.zero_()
            else:
                local.normal_(mean=0.0, std=std, generator=g)


def tp_param_ids(model_parts):
    """(tp group, ids of the tp-sharded parameters) of models that went through `apply_tp`, else (None, empty)."""
    group, ids = None, set()
    for m in model_parts:
        info = getattr(m, "_tn_tp", None)
        if info:
            group = info["group"]
            ids |= {id(p) for n, p in m.named_parameters() if n in info["sharded_names"]}
    return group, ids
