"""Tensor parallelism for the packed decoder blocks — BASELINE config E ("Kimi-Audio-7B … TP=2 × FSDP2 dp=4").

The reference's template is touchnet/models/llama/parallelize_llama.py:105-196 (`parallelize_module` with
Colwise / Rowwise / SequenceParallel plans over DTensors); it does not support Kimi-Audio at all.  Our
blocks do not call `nn.Linear.forward` (they hand weight tensors to multi-op HIP nodes), so the plan is applied
Megatron-style on the modules themselves, which is also what the kernels want (plain local tensors, no DTensor
dispatch on the hot path):

  attention   q / k / v projections COLUMN-parallel by heads (rank r owns query heads [r Nh/tp, (r+1) Nh/tp) and the
              kv heads they read — GQA groups never straddle ranks), o_proj ROW-parallel over the same heads
  MLP         gate / up column-parallel, down row-parallel
  per block   without sequence parallelism: ONE all-reduce of the attention output and ONE of the MLP output in forward
              (xGMI, RCCL), their mirrors (all-reduce of the input gradients) in backward; the residual stream, norms,
              embeddings and the heads stay replicated over the tp ranks
  sequence parallel (`sequence_parallel=True`; the reference's plan: embed Rowwise(out Shard(1)), norms SequenceParallel,
              o/down Rowwise(out Shard(1)), parallelize_llama.py:133-176): the residual stream, the fused add+norm kernels
              and the residual adds run on T/tp rows per rank; the all-reduce of a block output becomes a reduce-scatter
              along the sequence and the input of a column-parallel region an all-gather (same bytes on the wire, 1/tp of
              the norm / residual work and activation memory).  Norm weights see T/tp rows: their gradients are summed
              over the tp group after the backward (`reduce_sequence_partial_grads`, ONE flat all-reduce).
  loss parallel (`loss_parallel=True`; parallelize_llama.py:177-186 + touchnet/loss/cross_entropy.py:29-33): lm_head is
              sharded over the VOCABULARY; the fused lm_head + CE computes local logits, combines the row statistics over
              the tp group (max / sum-exp / target logit: three [rows] all-reduces per chunk) and back-propagates with the
              GLOBAL log-sum-exp through the unchanged CE kernels (a label outside the local shard simply has no local
              target); d(hidden) is summed over the group.

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


def tp_all_reduce(t: torch.Tensor, group, op=None) -> None:
    if getattr(group, "emulated", False):
        return
    dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=group)


def _tp_size(group) -> int:
    return group.size() if getattr(group, "emulated", False) else dist.get_world_size(group)


def _tp_rank(group) -> int:
    return group.get_local_rank() if getattr(group, "emulated", False) else dist.get_rank(group)


def tp_all_gather_seq(x: torch.Tensor, group) -> torch.Tensor:
    """[B, T/tp, ...] -> [B, T, ...] (rank r holds rows [r T/tp, (r+1) T/tp)).  An emulated rank tiles its own rows."""
    tp = _tp_size(group)
    if getattr(group, "emulated", False):
        return x.repeat(1, tp, *([1] * (x.dim() - 2)))
    x = x.contiguous()
    out = torch.empty((tp * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)   # rank-major concatenation
    dist.all_gather_into_tensor(out, x, group=group)
    out = out.view((tp,) + tuple(x.shape))
    if x.shape[0] == 1:
        return out.view((1, tp * x.shape[1]) + tuple(x.shape[2:]))
    return out.transpose(0, 1).reshape((x.shape[0], tp * x.shape[1]) + tuple(x.shape[2:]))


def tp_reduce_scatter_seq(x: torch.Tensor, group) -> torch.Tensor:
    """sum over the tp ranks of [B, T, ...], this rank keeps rows [r T/tp, (r+1) T/tp).  Emulated: the slice, unsummed."""
    tp, r = _tp_size(group), _tp_rank(group)
    B, T = x.shape[:2]
    if T % tp:
        raise ValueError(f"sequence length {T} is not divisible by tp = {tp}")
    if getattr(group, "emulated", False):
        return x.narrow(1, r * (T // tp), T // tp).contiguous()
    parts = x.reshape((B, tp, T // tp) + tuple(x.shape[2:])).transpose(0, 1).contiguous()       # [tp, B, T/tp, ...]
    out = torch.empty(parts.shape[1:], dtype=x.dtype, device=x.device)
    dist.reduce_scatter_tensor(out, parts.view((tp * B,) + tuple(parts.shape[2:])), group=group)
    return out


class _GatherSeq(torch.autograd.Function):
    """all-gather along the sequence; backward: `partial=True` -> reduce-scatter (what follows differs per rank: a
    column-parallel region, a vocabulary-parallel head), False -> this rank's slice (what follows is replicated)"""

    @staticmethod
    def forward(ctx, x, group, partial):
        ctx.group, ctx.partial = group, partial
        return tp_all_gather_seq(x, group)

    @staticmethod
    def backward(ctx, g):
        if ctx.partial:
            return tp_reduce_scatter_seq(g, ctx.group), None, None
        tp, r = _tp_size(ctx.group), _tp_rank(ctx.group)
        n = g.shape[1] // tp
        return g.narrow(1, r * n, n).contiguous(), None, None


class _ReduceScatterSeq(torch.autograd.Function):
    """sum over the tp ranks + scatter along the sequence (output of a row-parallel region); backward: all-gather"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return tp_reduce_scatter_seq(x, group)

    @staticmethod
    def backward(ctx, g):
        return tp_all_gather_seq(g, ctx.group), None


class _ScatterSeq(torch.autograd.Function):
    """this rank's rows of a replicated [B, T, ...] tensor (the embedding output); backward: all-gather, so that what
    produced the tensor — the replicated embedding — sees the gradient of every row on every rank"""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        tp, r = _tp_size(group), _tp_rank(group)
        if x.shape[1] % tp:
            raise ValueError(f"sequence length {x.shape[1]} is not divisible by tp = {tp}")
        n = x.shape[1] // tp
        return x.narrow(1, r * n, n).contiguous()

    @staticmethod
    def backward(ctx, g):
        return tp_all_gather_seq(g, ctx.group), None


class SequenceParallel:
    """What the decoder stacks consult (`model._tn_sp`): scatter behind the embedding, gather in front of the head."""

    def __init__(self, group, loss_parallel: bool):
        self.group, self.loss_parallel = group, loss_parallel

    def scatter(self, x):
        return _ScatterSeq.apply(x, self.group)

    def gather(self, h):
        return _GatherSeq.apply(h, self.group, self.loss_parallel)


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


def _wrap(module: nn.Module, group, sequence_parallel: bool = False) -> None:
    inner = module.forward

    if sequence_parallel:
        def forward(x, *args, **kwargs):
            return _ReduceScatterSeq.apply(inner(_GatherSeq.apply(x, group, True), *args, **kwargs), group)
    else:
        def forward(x, *args, **kwargs):
            return _ReduceFromTP.apply(inner(_CopyToTP.apply(x, group), *args, **kwargs), group)
    module.forward = forward


def _vocab_parallel(emb: nn.Embedding, group, rank: int) -> None:
    def forward(ids):
        vp = emb.weight.shape[0]
        local = ids - rank * vp
        outside = (local < 0) | (local >= vp)
        out = torch.nn.functional.embedding(local.clamp(0, vp - 1), emb.weight)
        out = out.masked_fill(outside.unsqueeze(-1), 0)
        return _ReduceFromTP.apply(out, group)              # forward: sum over tp; backward: every rank keeps the gradient
    emb.forward = forward


def _decoder_stacks(model: nn.Module):
    """the modules that own an embedding -> layers -> final norm loop (DecoderModel / KimiDecoderModel)"""
    lm = getattr(model, "language_model", model)
    return [lm.model]


def apply_tp(model: nn.Module, tp_mesh, loss_parallel: bool = False, sequence_parallel: bool = False) -> nn.Module:
    group, tp, rank = tp_mesh.get_group(), tp_mesh.size(), tp_mesh.get_local_rank()
    lm = getattr(model, "language_model", model)
    tied = bool(getattr(lm.config, "tie_word_embeddings", False)) and hasattr(lm, "lm_head") and any(
        getattr(st, "embed_tokens", None) is not None and st.embed_tokens.weight is lm.lm_head.weight
        for st in _decoder_stacks(model))
    if loss_parallel:
        if lm.lm_head.weight.shape[0] % tp:
            raise ValueError(f"vocabulary {lm.lm_head.weight.shape[0]} is not divisible by tp = {tp}")
        if not sequence_parallel:
            raise NotImplementedError("loss parallel is built on the sequence-parallel plan (as in the reference): pass "
                                      "sequence_parallel=True")
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
            _wrap(attn, group, sequence_parallel)
            _wrap(mlp, group, sequence_parallel)
    seq_partial = []
    if sequence_parallel:
        for stack in _decoder_stacks(model):
            stack._tn_sp = SequenceParallel(group, loss_parallel)
            for m in stack.modules():                     # every norm of the stack runs on T/tp rows of the sequence
                if type(m).__name__ == "RMSNorm":
                    seq_partial.append(m.weight)
    if loss_parallel:
        lm.lm_head.weight = _shard(lm.lm_head.weight, 0, rank, tp)
        sharded.append(lm.lm_head.weight)
        lm._tn_loss_parallel = (group, rank, tp)
        if tied:
            # Tied embeddings (Llama-3.2-1B, the reference's tiny test config): the ONE weight is vocabulary-sharded, so the
            # embedding becomes vocabulary-parallel as in the reference's plan (RowwiseParallel on `tok_embeddings`,
            # parallelize_llama.py:133-141): every rank looks up the ids of its own V/tp rows, the other rows are zero,
            # the sum over the tp ranks is the embedding.  (The sum is an all-reduce here, the sequence-parallel scatter
            # follows as for a replicated embedding: callers that add audio rows to the embedding output see a full tensor.)
            for st in _decoder_stacks(model):
                emb = st.embed_tokens
                emb.weight = lm.lm_head.weight
                emb.num_embeddings = emb.weight.shape[0]
                _vocab_parallel(emb, group, rank)
    ids, pids = {id(p) for p in sharded}, {id(p) for p in seq_partial}
    model._tn_tp = {"group": group, "size": tp, "rank": rank, "sequence_parallel": sequence_parallel,
                    "loss_parallel": loss_parallel,
                    "sharded_names": {n for n, p in model.named_parameters() if id(p) in ids},
                    "seq_partial_names": {n for n, p in model.named_parameters() if id(p) in pids}}
    _checkpoint_view(model, tp_mesh)
    return model


def _row_parallel(name: str) -> bool:
    return "o_proj" in name or "down_proj" in name


def _checkpoint_view(model: nn.Module, tp_mesh) -> None:
    """`state_dict()` hands the tensor-parallel shards out as DTensors on the tp mesh (Shard(1) for the row-parallel
    o_proj / down_proj weights, Shard(0) for everything else that is sharded), sharing storage with the parameters — what
    the reference's DTensor-based plan gives torch.distributed.checkpoint: an unsharded checkpoint loads into the shards,
    a sharded one reloads anywhere (the reference's tests/touchnet/models/test_llama.py).  `load_state_dict()` accepts the
    same view back (and full tensors): a pre-hook turns them into the local shards.  Only for a real DeviceMesh and
    plain (not FSDP2-wrapped) parameters: under tp x FSDP2 the dim-0 shards would need a strided 2-D placement."""
    try:
        from torch.distributed.device_mesh import DeviceMesh
        from torch.distributed.tensor import DTensor, Shard
    except Exception:                                        # pragma: no cover
        return
    if not isinstance(tp_mesh, DeviceMesh):
        return

    tp = tp_mesh.size()
    cache = {}

    def spmd_mesh(dp_mesh):
        """the 2-D (dp, tp) mesh a tp shard that FSDP2 sharded again lives on: rows = this tp column's dp ranks"""
        key = tuple(dp_mesh.mesh_dim_names or ())
        if key not in cache:
            root = tp_mesh._get_root_mesh() if hasattr(tp_mesh, "_get_root_mesh") else None
            try:
                cache[key] = root[key + ("tp",)]
            except Exception:
                cache[key] = None
        return cache[key]

    def hook(module, state_dict, prefix, local_metadata):
        names = module._tn_tp["sharded_names"]
        plain = lambda n: n.replace("_checkpoint_wrapped_module.", "")
        params = {plain(n): p for n, p in module.named_parameters(remove_duplicate=False)}
        shared = {id(params[n]) for n in names if n in params}
        for key in list(state_dict.keys()):
            n = plain(key[len(prefix):])
            p = params.get(n)
            t = state_dict[key]
            if p is None or id(p) not in shared or not isinstance(t, torch.Tensor) or t.is_meta:
                continue
            row = _row_parallel(n)
            if not isinstance(t, DTensor):
                state_dict[key] = DTensor.from_local(t, tp_mesh, [Shard(1 if row else 0)], run_check=False)
                continue
            # tp x FSDP2: FSDP2 sharded the tp-local tensor on dim 0 over its own (dp) mesh.  A row-parallel weight is
            # then Shard(0) over dp x Shard(1) over tp; a column-parallel one is split on dim 0 TWICE, tp-major — torch's
            # FSDP + TP convention for that is a strided shard on the dp dimension
            if len(t.placements) != 1 or not t.placements[0].is_shard(0) or t.shape[0] % t.device_mesh.size():
                continue
            mesh2d = spmd_mesh(t.device_mesh)
            if mesh2d is None:
                continue
            try:
                from torch.distributed.tensor.placement_types import _StridedShard
            except Exception:                                # pragma: no cover
                continue
            shape = list(t.shape)
            shape[1 if row else 0] *= tp
            placements = [Shard(0), Shard(1)] if row else [_StridedShard(0, split_factor=tp), Shard(0)]
            stride = [1] * len(shape)
            for d in range(len(shape) - 2, -1, -1):
                stride[d] = stride[d + 1] * shape[d + 1]
            state_dict[key] = DTensor.from_local(t.to_local(), mesh2d, placements, run_check=False,
                                                 shape=torch.Size(shape), stride=tuple(stride))
        return state_dict
    model._register_state_dict_hook(hook)

    def load_hook(module, state_dict, prefix, *unused):
        """The load side of the view (`model.load_state_dict(...)`; DCP loads in place and never gets here): an incoming
        DTensor — this model's own `state_dict()`, or another layout of the same global tensor — or a plain tensor of the
        GLOBAL shape becomes the local shard the plain parameter holds; local-shaped plain tensors pass through.  FSDP2-
        wrapped parameters (DTensors themselves) are left to FSDP2's own loading."""
        names = module._tn_tp["sharded_names"]
        rank = module._tn_tp["rank"]
        plain = lambda n: n.replace("_checkpoint_wrapped_module.", "")
        params = {plain(n): p for n, p in module.named_parameters(remove_duplicate=False)}
        shared = {id(params[n]) for n in names if n in params}
        for key in list(state_dict.keys()):
            n = plain(key[len(prefix):])
            p = params.get(n)
            t = state_dict[key]
            if p is None or id(p) not in shared or isinstance(p, DTensor) or not isinstance(t, torch.Tensor) or t.is_meta:
                continue
            dim = 1 if _row_parallel(n) else 0
            if isinstance(t, DTensor):
                same = t.device_mesh.ndim == 1 and t.device_mesh.size() == tp and tuple(t.device_mesh.mesh.flatten().tolist()) \
                    == tuple(tp_mesh.mesh.flatten().tolist())
                t = t.redistribute(placements=[Shard(dim)]).to_local() if same else t.full_tensor()
            if tuple(t.shape) != tuple(p.shape):
                want = list(p.shape)
                want[dim] *= tp
                if list(t.shape) != want:
                    continue                                  # (not ours to fix: load_state_dict reports the mismatch)
                t = t.narrow(dim, rank * p.shape[dim], p.shape[dim])
            state_dict[key] = t
    model._register_load_state_dict_pre_hook(load_hook, with_module=True)


@torch.no_grad()
def reduce_sequence_partial_grads(model: nn.Module) -> None:
    """Sequence parallelism: a norm weight saw T/tp rows per rank, its gradient is the sum over the tp group.  ONE flat
    all-reduce for all of them, after the backward (under FSDP2 on the local shards of the reduced gradients: the sum
    commutes with the dp average)."""
    info = getattr(model, "_tn_tp", None)
    if not info or not info.get("seq_partial_names"):
        return
    grads = []
    for n, p in model.named_parameters():
        if _plain_name(n) in info["seq_partial_names"] and p.grad is not None:
            g = p.grad
            grads.append(g._local_tensor if hasattr(g, "_local_tensor") else g)
    if not grads:
        return
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    tp_all_reduce(flat, info["group"])
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


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
        name = _plain_name(name)
        if name in info["sharded_names"] and not p.is_meta:
            local = p._local_tensor if hasattr(p, "_local_tensor") else p
            if name.endswith("bias"):
                local.zero_()
                continue
            g = torch.Generator(device=local.device)
            g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode()) * 31 + info["rank"]) % (2 ** 63 - 1))
            if hasattr(p, "_local_tensor"):
                # FSDP2 on top (dim-0 shards over dp): every dp rank draws the WHOLE tp-local tensor from the same
                # (seed, name, tp rank) stream and keeps its own rows — drawing `local` directly would give all dp
                # ranks the same numbers, i.e. dp copies of one row block
                from torch.distributed.tensor._utils import compute_local_shape_and_global_offset
                shape, offset = compute_local_shape_and_global_offset(p.shape, p.device_mesh, p.placements)
                full = torch.empty(tuple(p.shape), dtype=local.dtype, device=local.device)
                full.normal_(mean=0.0, std=std, generator=g)
                idx = tuple(slice(o, o + n) for o, n in zip(offset, shape))
                local.copy_(full[idx].reshape(local.shape))
            else:
                local.normal_(mean=0.0, std=std, generator=g)


def _plain_name(name: str) -> str:
    """Parameter name without the prefix activation checkpointing's wrapper inserts (`apply_ac` runs after `apply_tp`,
    which recorded the names of the unwrapped blocks)."""
    return name.replace("_checkpoint_wrapped_module.", "")


def tp_param_ids(model_parts):
    """(tp group, ids of the tp-sharded parameters) of models that went through `apply_tp`, else (None, empty)."""
    group, ids = None, set()
    for m in model_parts:
        info = getattr(m, "_tn_tp", None)
        if info:
            group = info["group"]
            ids |= {id(p) for n, p in m.named_parameters() if _plain_name(n) in info["sharded_names"]}
    return group, ids
