"""PyTorch-ROCm custom ops `torch.ops.mi355_touch.*` over the C ABI of libtouchnet_amd.so (SURVEY §8b "C-ABI realisation").

Each op has a schema (inferred from the annotations), a device implementation that only ENQUEUES the HIP kernel on the
current stream (outputs come from the caching allocator; no sync), a `register_fake` meta kernel (shapes / dtypes only:
meta-device construction, FakeTensor tracing, `torch.compile` of the surrounding module, activation-checkpoint
policies that list ops — touchnet/models/helper_func.py:39-96 — all see the ops without executing them) and a
`register_autograd` formula whose backward is again an op of this namespace.  `touchnet_amd.functional` is the
user-facing layer on top (it adds argument normalisation and the multi-op autograd nodes of the linear layers).

Conventions: activations are [..., H] dense; "rows" = product of the leading dims.  No op returns an alias of an input
(the dispatcher forbids it): where the kernel would pass an input through (the residual stream without an add), the op
returns an EMPTY placeholder and the autograd formula keeps the input itself.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import _C

NS = "mi355_touch"
_p, _cur = _C.ptr, _C.stream


def _lib():
    return _C.lib()


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _empty0(like: Tensor) -> Tensor:
    return like.new_empty(0)


# ===================================================================================================== RMSNorm
@custom_op(f"{NS}::rmsnorm_fwd", mutates_args=(), device_types="cuda")
def rmsnorm_fwd(x: Tensor, residual: Optional[Tensor], weight: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (y, h, rstd): h = x + residual (EMPTY when residual is None: then h == x), y = w * T(h * rstd)."""
    H = x.shape[-1]
    x2 = _c(x).view(-1, H)
    rows = x2.shape[0]
    r2 = _c(residual).view(-1, H) if residual is not None else None
    w = _c(weight).to(x.dtype)
    y = torch.empty_like(x2)
    h = torch.empty_like(x2) if r2 is not None else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _C.check(_lib().tn_rmsnorm_fwd(_p(x2), _p(r2), _p(w), _p(y), _p(h), _p(rstd), rows, H, float(eps), _C.dcode(x2),
                                   _cur()), "tn_rmsnorm_fwd")
    return y.view(x.shape), (h.view(x.shape) if h is not None else _empty0(x)), rstd


@rmsnorm_fwd.register_fake
def _(x, residual, weight, eps):
    rows = x.numel() // x.shape[-1]
    return (torch.empty_like(x, memory_format=torch.contiguous_format),
            torch.empty_like(x, memory_format=torch.contiguous_format) if residual is not None else x.new_empty(0),
            x.new_empty(rows, dtype=torch.float32))


@custom_op(f"{NS}::rmsnorm_bwd", mutates_args=(), device_types="cuda")
def rmsnorm_bwd(dy: Tensor, h: Tensor, weight: Tensor, rstd: Tensor, dres: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """-> (dh, dw): dh includes the incoming residual-stream gradient `dres`."""
    H = h.shape[-1]
    h2 = _c(h).view(-1, H)
    rows = h2.shape[0]
    dy2 = _c(dy).view(rows, H)
    dr2 = _c(dres).view(rows, H) if dres is not None else None
    w = _c(weight).to(h.dtype)
    dh = torch.empty_like(h2)
    dw = torch.empty(H, dtype=h.dtype, device=h.device)
    ws = torch.empty(_lib().tn_norm_bwd_workspace_floats(rows, H), dtype=torch.float32, device=h.device)
    _C.check(_lib().tn_rmsnorm_bwd(_p(dy2), _p(h2), _p(w), _p(rstd), _p(dr2), _p(dh), _p(dw), _p(ws), rows, H,
                                   _C.dcode(h2), _cur()), "tn_rmsnorm_bwd")
    return dh.view(h.shape), dw


@rmsnorm_bwd.register_fake
def _(dy, h, weight, rstd, dres):
    return torch.empty_like(h, memory_format=torch.contiguous_format), h.new_empty(h.shape[-1])


def _rmsnorm_setup(ctx, inputs, output):
    x, residual, weight, eps = inputs
    _, h, rstd = output
    ctx.has_res = residual is not None
    ctx.wdtype = weight.dtype
    ctx.set_materialize_grads(False)     # (no zero tensors for the statistics outputs' gradients: 2-3 tiny fills per norm)
    ctx.save_for_backward(h if ctx.has_res else x, weight, rstd)


def _rmsnorm_backward(ctx, dy, dh_new, _drstd):
    h, weight, rstd = ctx.saved_tensors
    dres = dh_new if (ctx.has_res and dh_new is not None) else None
    if dy is None:
        dy = torch.zeros_like(h)
    dh, dw = rmsnorm_bwd(dy, h, weight, rstd, dres)
    return dh, (dh if ctx.has_res else None), dw.to(ctx.wdtype), None


rmsnorm_fwd.register_autograd(_rmsnorm_backward, setup_context=_rmsnorm_setup)


# ===================================================================================================== LayerNorm
@custom_op(f"{NS}::layernorm_fwd", mutates_args=(), device_types="cuda")
def layernorm_fwd(x: Tensor, residual: Optional[Tensor], weight: Tensor, bias: Tensor,
                  eps: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (y, h, mean, rstd); h EMPTY when residual is None."""
    H = x.shape[-1]
    x2 = _c(x).view(-1, H)
    rows = x2.shape[0]
    r2 = _c(residual).view(-1, H) if residual is not None else None
    w, b = _c(weight).to(x.dtype), _c(bias).to(x.dtype)
    y = torch.empty_like(x2)
    h = torch.empty_like(x2) if r2 is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    _C.check(_lib().tn_layernorm_fwd(_p(x2), _p(r2), _p(w), _p(b), _p(y), _p(h), _p(mean), _p(rstd), rows, H,
                                     float(eps), _C.dcode(x2), _cur()), "tn_layernorm_fwd")
    return y.view(x.shape), (h.view(x.shape) if h is not None else _empty0(x)), mean, rstd


@layernorm_fwd.register_fake
def _(x, residual, weight, bias, eps):
    rows = x.numel() // x.shape[-1]
    e = lambda: torch.empty_like(x, memory_format=torch.contiguous_format)
    return (e(), e() if residual is not None else x.new_empty(0), x.new_empty(rows, dtype=torch.float32),
            x.new_empty(rows, dtype=torch.float32))


@custom_op(f"{NS}::layernorm_bwd", mutates_args=(), device_types="cuda")
def layernorm_bwd(dy: Tensor, h: Tensor, weight: Tensor, mean: Tensor, rstd: Tensor,
                  dres: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    H = h.shape[-1]
    h2 = _c(h).view(-1, H)
    rows = h2.shape[0]
    dy2 = _c(dy).view(rows, H)
    dr2 = _c(dres).view(rows, H) if dres is not None else None
    w = _c(weight).to(h.dtype)
    dh = torch.empty_like(h2)
    dw = torch.empty(H, dtype=h.dtype, device=h.device)
    db = torch.empty_like(dw)
    ws = torch.empty(_lib().tn_norm_bwd_workspace_floats(rows, H), dtype=torch.float32, device=h.device)
    _C.check(_lib().tn_layernorm_bwd(_p(dy2), _p(h2), _p(w), _p(mean), _p(rstd), _p(dr2), _p(dh), _p(dw), _p(db),
                                     _p(ws), rows, H, _C.dcode(h2), _cur()), "tn_layernorm_bwd")
    return dh.view(h.shape), dw, db


@layernorm_bwd.register_fake
def _(dy, h, weight, mean, rstd, dres):
    return (torch.empty_like(h, memory_format=torch.contiguous_format), h.new_empty(h.shape[-1]),
            h.new_empty(h.shape[-1]))


def _layernorm_setup(ctx, inputs, output):
    x, residual, weight, bias, eps = inputs
    _, h, mean, rstd = output
    ctx.has_res = residual is not None
    ctx.wdtype = weight.dtype
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(h if ctx.has_res else x, weight, mean, rstd)


def _layernorm_backward(ctx, dy, dh_new, _dm, _dr):
    h, weight, mean, rstd = ctx.saved_tensors
    dres = dh_new if (ctx.has_res and dh_new is not None) else None
    if dy is None:
        dy = torch.zeros_like(h)
    dh, dw, db = layernorm_bwd(dy, h, weight, mean, rstd, dres)
    return dh, (dh if ctx.has_res else None), dw.to(ctx.wdtype), db.to(ctx.wdtype), None


layernorm_fwd.register_autograd(_layernorm_backward, setup_context=_layernorm_setup)


# ===================================================================================================== activations
@custom_op(f"{NS}::swiglu_fwd", mutates_args=(), device_types="cuda")
def swiglu_fwd(gate: Tensor, up: Tensor) -> Tensor:
    g, u = _c(gate), _c(up)
    out = torch.empty_like(g)
    _C.check(_lib().tn_swiglu_fwd(_p(g), _p(u), _p(out), g.numel(), _C.dcode(g), _cur()), "tn_swiglu_fwd")
    return out


@swiglu_fwd.register_fake
def _(gate, up):
    return torch.empty_like(gate, memory_format=torch.contiguous_format)


@custom_op(f"{NS}::swiglu_bwd", mutates_args=(), device_types="cuda")
def swiglu_bwd(dout: Tensor, gate: Tensor, up: Tensor) -> Tuple[Tensor, Tensor]:
    d, g, u = _c(dout), _c(gate), _c(up)
    dg, du = torch.empty_like(g), torch.empty_like(u)
    _C.check(_lib().tn_swiglu_bwd(_p(d), _p(g), _p(u), _p(dg), _p(du), g.numel(), _C.dcode(g), _cur()),
             "tn_swiglu_bwd")
    return dg, du


@swiglu_bwd.register_fake
def _(dout, gate, up):
    e = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return e(gate), e(up)


swiglu_fwd.register_autograd(lambda ctx, d: swiglu_bwd(d, *ctx.saved_tensors),
                             setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


@custom_op(f"{NS}::gelu_fwd", mutates_args=(), device_types="cuda")
def gelu_fwd(x: Tensor) -> Tensor:
    x = _c(x)
    out = torch.empty_like(x)
    _C.check(_lib().tn_gelu_fwd(_p(x), _p(out), x.numel(), _C.dcode(x), _cur()), "tn_gelu_fwd")
    return out


@gelu_fwd.register_fake
def _(x):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@custom_op(f"{NS}::gelu_bwd", mutates_args=(), device_types="cuda")
def gelu_bwd(dout: Tensor, x: Tensor) -> Tensor:
    d, x = _c(dout), _c(x)
    dx = torch.empty_like(x)
    _C.check(_lib().tn_gelu_bwd(_p(d), _p(x), _p(dx), x.numel(), _C.dcode(x), _cur()), "tn_gelu_bwd")
    return dx


@gelu_bwd.register_fake
def _(dout, x):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


gelu_fwd.register_autograd(lambda ctx, d: gelu_bwd(d, *ctx.saved_tensors),
                           setup_context=lambda ctx, inputs, output: ctx.save_for_backward(*inputs))


# ===================================================================================================== RoPE
@custom_op(f"{NS}::rope_apply", mutates_args=(), device_types="cuda")
def rope_apply(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor, backward: bool) -> Tuple[Tensor, Tensor]:
    """q [B,T,Nh,D], k [B,T,Nkv,D]; cos/sin [B*T, D/2].  `backward` applies the transposed rotation (gradients)."""
    q, k = _c(q), _c(k)
    B, T, hq, D = q.shape
    hk = k.shape[2]
    qo, ko = torch.empty_like(q), torch.empty_like(k)
    _C.check(_lib().tn_rope_apply(_p(q), _p(k), _p(qo), _p(ko), _p(cos), _p(sin), B * T, hq, hk, D, int(backward),
                                  _C.dcode(q), _cur()), "tn_rope_apply")
    return qo, ko


@rope_apply.register_fake
def _(q, k, cos, sin, backward):
    e = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return e(q), e(k)


def _rope_setup(ctx, inputs, output):
    _, _, cos, sin, backward = inputs
    ctx.save_for_backward(cos, sin)
    ctx.backward_flag = backward


def _rope_backward(ctx, dq, dk):
    cos, sin = ctx.saved_tensors
    gq, gk = rope_apply(dq, dk, cos, sin, not ctx.backward_flag)
    return gq, gk, None, None, None


rope_apply.register_autograd(_rope_backward, setup_context=_rope_setup)


# ===================================================================================================== attention
@custom_op(f"{NS}::attn_fwd", mutates_args=(), device_types="cuda")
def attn_fwd(q: Tensor, k: Tensor, v: Tensor, doc: Tensor, meta: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
    """Packed (document-masked causal) attention: q [B,T,Nh,D], k/v [B,T,Nkv,D] bf16, doc int32 [B,T], meta from
    `attn_build_meta` -> (o [B,T,Nh,D], lse2 fp32 [B,Nh,T])."""
    q, k, v = _c(q), _c(k), _c(v)
    if q.dtype != torch.bfloat16:
        raise _C.KernelError("packed_attention: bf16 only (MFMA 32x32x16 bf16 kernel)")
    B, T, Nh, D = q.shape
    Nkv = k.shape[2]
    if tuple(doc.shape) != (B, T):
        raise _C.KernelError(f"mask built for {tuple(doc.shape)}, got q {(B, T)}")
    o = torch.empty_like(q)
    lse2 = torch.empty(B, Nh, T, dtype=torch.float32, device=q.device)
    _C.check(_lib().tn_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse2), _p(doc), _p(meta), B, T, Nh, Nkv, D,
                                float(scale), _cur()), "tn_attn_fwd")
    return o, lse2


@attn_fwd.register_fake
def _(q, k, v, doc, meta, scale):
    B, T, Nh, D = q.shape
    return torch.empty_like(q, memory_format=torch.contiguous_format), q.new_empty(B, Nh, T, dtype=torch.float32)


@custom_op(f"{NS}::attn_bwd", mutates_args=(), device_types="cuda")
def attn_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse2: Tensor, doc: Tensor, meta: Tensor,
             scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    do = _c(do)
    B, T, Nh, D = q.shape
    Nkv = k.shape[2]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse2)
    _C.check(_lib().tn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse2), _p(delta), _p(dq), _p(dk), _p(dv),
                                _p(doc), _p(meta), B, T, Nh, Nkv, D, float(scale), _cur()), "tn_attn_bwd")
    return dq, dk, dv


@attn_bwd.register_fake
def _(q, k, v, o, do, lse2, doc, meta, scale):
    e = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return e(q), e(k), e(v)


@custom_op(f"{NS}::attn_fwd_bidir", mutates_args=(), device_types="cuda")
def attn_fwd_bidir(q: Tensor, k: Tensor, v: Tensor, doc: Tensor, meta: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
    """`attn_fwd` with the mask bidirectional inside a document (tn_attn_fwd_bidir)."""
    q, k, v = _c(q), _c(k), _c(v)
    if q.dtype != torch.bfloat16:
        raise _C.KernelError("bidirectional_attention: bf16 only (MFMA 32x32x16 bf16 kernel)")
    B, T, Nh, D = q.shape
    Nkv = k.shape[2]
    if tuple(doc.shape) != (B, T):
        raise _C.KernelError(f"mask built for {tuple(doc.shape)}, got q {(B, T)}")
    o = torch.empty_like(q)
    lse2 = torch.empty(B, Nh, T, dtype=torch.float32, device=q.device)
    _C.check(_lib().tn_attn_fwd_bidir(_p(q), _p(k), _p(v), _p(o), _p(lse2), _p(doc), _p(meta), B, T, Nh, Nkv, D,
                                      float(scale), _cur()), "tn_attn_fwd_bidir")
    return o, lse2


@attn_fwd_bidir.register_fake
def _(q, k, v, doc, meta, scale):
    B, T, Nh, D = q.shape
    return torch.empty_like(q, memory_format=torch.contiguous_format), q.new_empty(B, Nh, T, dtype=torch.float32)


@custom_op(f"{NS}::attn_bwd_bidir", mutates_args=(), device_types="cuda")
def attn_bwd_bidir(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse2: Tensor, doc: Tensor, meta: Tensor,
                   scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    do = _c(do)
    B, T, Nh, D = q.shape
    Nkv = k.shape[2]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse2)
    _C.check(_lib().tn_attn_bwd_bidir(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse2), _p(delta), _p(dq), _p(dk), _p(dv),
                                      _p(doc), _p(meta), B, T, Nh, Nkv, D, float(scale), _cur()), "tn_attn_bwd_bidir")
    return dq, dk, dv


@attn_bwd_bidir.register_fake
def _(q, k, v, o, do, lse2, doc, meta, scale):
    e = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return e(q), e(k), e(v)


_STACKED_BWD = os.environ.get("TN_ATTN_BWD_STACKED", "1") != "0"      # (A/B switch)


@custom_op(f"{NS}::attn_bwd_stacked", mutates_args=(), device_types="cuda")
def attn_bwd_stacked(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse2: Tensor, doc: Tensor, meta: Tensor,
                     scale: float) -> Tensor:
    """attn_bwd for Nh == Nkv with the three gradients in ONE buffer [3, B, T, Nh, D] (dq, dk, dv = its slices): a
    consumer that wants them side by side — the fused q/k/v weight gradient of the audio tower — can then run a batched
    GEMM over the buffer instead of concatenating three tensors first (functional._LinearGroup, wgrad="nt_fused")."""
    do = _c(do)
    B, T, Nh, D = q.shape
    out = q.new_empty(3, B, T, Nh, D)
    delta = torch.empty_like(lse2)
    _C.check(_lib().tn_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse2), _p(delta), _p(out[0]), _p(out[1]),
                                _p(out[2]), _p(doc), _p(meta), B, T, Nh, Nh, D, float(scale), _cur()), "tn_attn_bwd")
    return out


@attn_bwd_stacked.register_fake
def _(q, k, v, o, do, lse2, doc, meta, scale):
    B, T, Nh, D = q.shape
    return q.new_empty(3, B, T, Nh, D)


@custom_op(f"{NS}::attn_bwd_rope", mutates_args=(), device_types="cuda")
def attn_bwd_rope(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse2: Tensor, doc: Tensor, meta: Tensor,
                  scale: float, cos: Tensor, sin: Tensor, stacked: bool) -> Tensor:
    """attn_bwd for ROTATED q / k whose gradients are wanted for the un-rotated projections (tn_attn_bwd_rope: the rotary
    embedding's backward in the attention kernels' epilogues; bit-identical to attn_bwd followed by rope_apply(backward)).
    One buffer: stacked (Nh == Nkv) [3, B, T, Nh, D] = dq, dk, dv; otherwise flat [B * T * (Nh + 2 Nkv) * D] = dq | dk | dv."""
    do = _c(do)
    B, T, Nh, D = q.shape
    Nkv = k.shape[2]
    if not (cos.dtype == torch.bfloat16 and sin.dtype == torch.bfloat16 and cos.is_contiguous() and sin.is_contiguous()
            and cos.numel() == B * T * (D // 2) and sin.numel() == cos.numel()):
        raise _C.KernelError("attn_bwd_rope: cos / sin are the bf16 [B * T, head_dim / 2] tables of rope_tables")
    if stacked:
        out = q.new_empty(3, B, T, Nh, D)
        dq, dk, dv = out[0], out[1], out[2]
    else:
        out = q.new_empty(B * T * (Nh + 2 * Nkv) * D)
        dq, dk, dv = out.split([B * T * Nh * D, B * T * Nkv * D, B * T * Nkv * D])
    delta = torch.empty_like(lse2)
    _C.check(_lib().tn_attn_bwd_rope(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse2), _p(delta), _p(dq), _p(dk), _p(dv),
                                     _p(doc), _p(meta), B, T, Nh, Nkv, D, float(scale), _p(cos), _p(sin), _cur()),
             "tn_attn_bwd_rope")
    return out


@attn_bwd_rope.register_fake
def _(q, k, v, o, do, lse2, doc, meta, scale, cos, sin, stacked):
    B, T, Nh, D = q.shape
    return q.new_empty(3, B, T, Nh, D) if stacked else q.new_empty(B * T * (Nh + 2 * k.shape[2]) * D)


def _attn_setup(ctx, inputs, output):
    q, k, v, doc, meta, scale = inputs
    o, lse2 = output
    ctx.save_for_backward(_c(q), _c(k), _c(v), o, lse2, doc, meta)
    ctx.scale = scale
    ctx.set_materialize_grads(False)     # (the LSE output has no gradient: no [B, Nh, T] zero fill per layer)


def _attn_backward(ctx, do, _dlse):
    q, k, v, o, lse2, doc, meta = ctx.saved_tensors
    if do is None:
        do = torch.zeros_like(o)
    if q.shape == k.shape and _STACKED_BWD:                  # multi-head attention: one buffer, three slices
        dq, dk, dv = attn_bwd_stacked(q, k, v, o, do, lse2, doc, meta, ctx.scale).unbind(0)
    else:
        dq, dk, dv = attn_bwd(q, k, v, o, do, lse2, doc, meta, ctx.scale)
    return dq, dk, dv, None, None, None


attn_fwd.register_autograd(_attn_backward, setup_context=_attn_setup)


@custom_op(f"{NS}::attn_build_meta", mutates_args=(), device_types="cuda")
def attn_build_meta(doc: Tensor) -> Tensor:
    """doc int32 [B, T] -> per-64-tile range metadata (int32 [tn_attn_meta_ints(B, T)]); once per batch."""
    B, T = doc.shape
    meta = torch.empty(_lib().tn_attn_meta_ints(B, T), dtype=torch.int32, device=doc.device)
    _C.check(_lib().tn_attn_build_meta(_p(doc), _p(meta), B, T, _cur()), "tn_attn_build_meta")
    return meta


@attn_build_meta.register_fake
def _(doc):
    B, T = doc.shape
    # (tn_attn_meta_ints: tile statistics + per-wave statistics + precomputed KV-tile lists, csrc/attn_common.h)
    return doc.new_empty(5 * B * ((T + 63) // 64) + 4 * B * ((T + 31) // 32) + 520 * B * ((T + 127) // 128),
                         dtype=torch.int32)


def _segs_array(segs: List[int]):
    import ctypes
    flat = list(segs) + [0] * (6 - len(segs))
    return (ctypes.c_int * 6)(*flat)


@custom_op(f"{NS}::attn_fwd_seg", mutates_args=(), device_types="cuda")
def attn_fwd_seg(q: Tensor, k: Tensor, v: Tensor, doc: Tensor, meta: Tensor, scale: float, segs: List[int],
                 rows_per_batch: int) -> Tuple[Tensor, Tensor]:
    """Sequence-sharded query side (context parallelism): q [B, R, Nh, D] local rows, k/v [B, T, Nkv, D] global;
    `segs` = up to two (row0, rows, global_offset) triples, flattened."""
    q, k, v = _c(q), _c(k), _c(v)
    if q.dtype != torch.bfloat16:
        raise _C.KernelError("packed_attention_sharded: bf16 only")
    B, R, Nh, D = q.shape
    T, Nkv = k.shape[1], k.shape[2]
    if tuple(doc.shape) != (B, T) or R != rows_per_batch:
        raise _C.KernelError(f"mask {tuple(doc.shape)} / shard {rows_per_batch} vs q {(B, R)} k {(B, T)}")
    o = torch.empty_like(q)
    lse2 = torch.empty(B, Nh, R, dtype=torch.float32, device=q.device)
    _C.check(_lib().tn_attn_fwd_seg(_p(q), _p(k), _p(v), _p(o), _p(lse2), _p(doc), _p(meta), B, T, Nh, Nkv, D,
                                    float(scale), len(segs) // 3, _segs_array(segs), R, _cur()), "tn_attn_fwd_seg")
    return o, lse2


@attn_fwd_seg.register_fake
def _(q, k, v, doc, meta, scale, segs, rows_per_batch):
    B, R, Nh, D = q.shape
    return torch.empty_like(q, memory_format=torch.contiguous_format), q.new_empty(B, Nh, R, dtype=torch.float32)


@custom_op(f"{NS}::attn_fwd_seg_chunks", mutates_args=(), device_types="cuda")
def attn_fwd_seg_chunks(q: Tensor, k: Tensor, v: Tensor, doc: Tensor, meta: Tensor, scale: float, segs: List[int],
                        rows_per_batch: int, chunk_len: int, chunk_mask: int) -> Tuple[Tensor, Tensor]:
    """attn_fwd_seg with the KEY side restricted to the sequence chunks whose bit is set in `chunk_mask` (chunk c =
    positions [c * chunk_len, (c + 1) * chunk_len)).  Rows without a key there: O = 0, LSE2 = +inf.  One half of the
    local / remote split of context-parallel attention (no autograd formula of its own: functional._SplitAttention
    differentiates the MERGED result with one attn_bwd_seg)."""
    q, k, v = _c(q), _c(k), _c(v)
    if q.dtype != torch.bfloat16:
        raise _C.KernelError("packed_attention_sharded: bf16 only")
    B, R, Nh, D = q.shape
    T, Nkv = k.shape[1], k.shape[2]
    if tuple(doc.shape) != (B, T) or R != rows_per_batch:
        raise _C.KernelError(f"mask {tuple(doc.shape)} / shard {rows_per_batch} vs q {(B, R)} k {(B, T)}")
    o = torch.empty_like(q)
    lse2 = torch.empty(B, Nh, R, dtype=torch.float32, device=q.device)
    _C.check(_lib().tn_attn_fwd_seg_chunks(_p(q), _p(k), _p(v), _p(o), _p(lse2), _p(doc), _p(meta), B, T, Nh, Nkv, D,
                                           float(scale), len(segs) // 3, _segs_array(segs), R, int(chunk_len),
                                           int(chunk_mask), _cur()), "tn_attn_fwd_seg_chunks")
    return o, lse2


@attn_fwd_seg_chunks.register_fake
def _(q, k, v, doc, meta, scale, segs, rows_per_batch, chunk_len, chunk_mask):
    B, R, Nh, D = q.shape
    return torch.empty_like(q, memory_format=torch.contiguous_format), q.new_empty(B, Nh, R, dtype=torch.float32)


@custom_op(f"{NS}::attn_merge", mutates_args=(), device_types="cuda")
def attn_merge(o_a: Tensor, lse2_a: Tensor, o_b: Tensor, lse2_b: Tensor) -> Tuple[Tensor, Tensor]:
    """Two partial attention results over disjoint key sets -> (O, LSE2) of their union (log2-domain LSE, +inf = empty)."""
    o_a, o_b, la, lb = _c(o_a), _c(o_b), _c(lse2_a), _c(lse2_b)
    B, R, Nh, D = o_a.shape
    o, lse2 = torch.empty_like(o_a), torch.empty_like(la)
    _C.check(_lib().tn_attn_merge(_p(o_a), _p(la), _p(o_b), _p(lb), _p(o), _p(lse2), B, R, Nh, D, _cur()),
             "tn_attn_merge")
    return o, lse2


@attn_merge.register_fake
def _(o_a, lse2_a, o_b, lse2_b):
    return (torch.empty_like(o_a, memory_format=torch.contiguous_format),
            torch.empty_like(lse2_a, memory_format=torch.contiguous_format))


@custom_op(f"{NS}::attn_bwd_seg", mutates_args=(), device_types="cuda")
def attn_bwd_seg(q: Tensor, k: Tensor, v: Tensor, o: Tensor, do: Tensor, lse2: Tensor, doc: Tensor, meta: Tensor,
                 scale: float, segs: List[int], rows_per_batch: int) -> Tuple[Tensor, Tensor, Tensor]:
    """dk / dv are this rank's PARTIAL sums over the global [B, T, Nkv, D]."""
    do = _c(do)
    B, R, Nh, D = q.shape
    T, Nkv = k.shape[1], k.shape[2]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty_like(lse2)
    _C.check(_lib().tn_attn_bwd_seg(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse2), _p(delta), _p(dq), _p(dk), _p(dv),
                                    _p(doc), _p(meta), B, T, Nh, Nkv, D, float(scale), len(segs) // 3,
                                    _segs_array(segs), R, _cur()), "tn_attn_bwd_seg")
    return dq, dk, dv


@attn_bwd_seg.register_fake
def _(q, k, v, o, do, lse2, doc, meta, scale, segs, rows_per_batch):
    e = lambda t: torch.empty_like(t, memory_format=torch.contiguous_format)
    return e(q), e(k), e(v)


def _attn_seg_setup(ctx, inputs, output):
    q, k, v, doc, meta, scale, segs, rpb = inputs
    o, lse2 = output
    ctx.save_for_backward(_c(q), _c(k), _c(v), o, lse2, doc, meta)
    ctx.scale, ctx.segs, ctx.rpb = scale, list(segs), rpb
    ctx.set_materialize_grads(False)


def _attn_seg_backward(ctx, do, _dlse):
    q, k, v, o, lse2, doc, meta = ctx.saved_tensors
    if do is None:
        do = torch.zeros_like(o)
    dq, dk, dv = attn_bwd_seg(q, k, v, o, do, lse2, doc, meta, ctx.scale, ctx.segs, ctx.rpb)
    return dq, dk, dv, None, None, None, None, None


attn_fwd_seg.register_autograd(_attn_seg_backward, setup_context=_attn_seg_setup)


# ===================================================================================================== cross-entropy
@custom_op(f"{NS}::ce_fwd", mutates_args=(), device_types="cuda")
def ce_fwd(logits: Tensor, labels: Tensor, sentence_lens: Tensor, num_sentence: Tensor,
           ignore_index: int) -> Tuple[Tensor, Tensor, Tensor]:
    """logits [n, V]; labels / sentence_lens int64 [n]; num_sentence fp32 [1] ->
    (loss_per_sample 0-d [differentiable], stats fp32 [4] = {per_sample, per_token, accuracy, n_valid}, lse fp32 [n])."""
    n, V = logits.shape
    nll = torch.empty(n, dtype=torch.float32, device=logits.device)
    lse = torch.empty_like(nll)
    hit = torch.empty(n, dtype=torch.int32, device=logits.device)
    out = torch.empty(4, dtype=torch.float32, device=logits.device)
    _C.check(_lib().tn_ce_forward(_p(logits), _p(labels), _p(sentence_lens), _p(num_sentence), _p(nll), _p(lse),
                                  _p(hit), _p(out), n, V, int(ignore_index), _C.dcode(logits), _cur()),
             "tn_ce_forward")
    return out[0].clone(), out, lse


@ce_fwd.register_fake
def _(logits, labels, sentence_lens, num_sentence, ignore_index):
    f32 = dict(dtype=torch.float32)
    return logits.new_empty((), **f32), logits.new_empty(4, **f32), logits.new_empty(logits.shape[0], **f32)


@custom_op(f"{NS}::ce_bwd", mutates_args=(), device_types="cuda")
def ce_bwd(logits: Tensor, labels: Tensor, sentence_lens: Tensor, lse: Tensor, num_sentence: Tensor, grad_out: Tensor,
           ignore_index: int) -> Tensor:
    n, V = logits.shape
    dlog = torch.empty_like(logits)
    _C.check(_lib().tn_ce_backward(_p(logits), _p(dlog), _p(labels), _p(sentence_lens), _p(lse), _p(num_sentence),
                                   _p(grad_out), n, V, int(ignore_index), _C.dcode(logits), _cur()), "tn_ce_backward")
    return dlog


@ce_bwd.register_fake
def _(logits, labels, sentence_lens, lse, num_sentence, grad_out, ignore_index):
    return torch.empty_like(logits)


@custom_op(f"{NS}::ce_bwd_", mutates_args=("logits",), device_types="cuda")
def ce_bwd_(logits: Tensor, labels: Tensor, sentence_lens: Tensor, lse: Tensor, num_sentence: Tensor,
            grad_out: Tensor, ignore_index: int) -> None:
    """In-place form: dlogits overwrite the logits (they are dead after the loss: saves n x V x 2 bytes)."""
    n, V = logits.shape
    _C.check(_lib().tn_ce_backward(_p(logits), _p(logits), _p(labels), _p(sentence_lens), _p(lse), _p(num_sentence),
                                   _p(grad_out), n, V, int(ignore_index), _C.dcode(logits), _cur()), "tn_ce_backward")


def _ce_setup(ctx, inputs, output):
    logits, labels, sentence_lens, num_sentence, ignore_index = inputs
    _, _, lse = output
    ctx.save_for_backward(logits, labels, sentence_lens, lse, num_sentence)
    ctx.ignore_index = ignore_index


def _ce_backward(ctx, g_loss, _g_stats, _g_lse):
    logits, labels, sentence_lens, lse, ns = ctx.saved_tensors
    g = _c(g_loss).to(torch.float32).reshape(1)
    return ce_bwd(logits, labels, sentence_lens, lse, ns, g, ctx.ignore_index), None, None, None, None


ce_fwd.register_autograd(_ce_backward, setup_context=_ce_setup)


# ===================================================================================================== GEMM
@custom_op(f"{NS}::gemm_tn", mutates_args=(), device_types="cuda")
def gemm_tn(a: Tensor, b: Tensor, bias: Optional[Tensor]) -> Tensor:
    """a [M, K] · b [N, K]^T (+ bias [N]) -> [M, N]; bf16, fp32 accumulation — the hand-written MFMA kernel
    (csrc/gemm.hip).  Differentiable: both gradients are GEMMs of the same kernel on transposed operands."""
    if a.dim() != 2 or b.dim() != 2 or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise _C.KernelError("gemm_tn: 2-D bf16 operands only")
    a, b = (a if a.stride(1) == 1 else a.contiguous()), (b if b.stride(1) == 1 else b.contiguous())
    M, K = a.shape
    N = b.shape[0]
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    bias_c = _c(bias).to(a.dtype) if bias is not None else None
    _C.check(_lib().tn_gemm_bf16_tn(_p(a), _p(b), _p(out), None, _p(bias_c), M, N, K, a.stride(0), b.stride(0), N, 0,
                                    0, _cur()), "tn_gemm_bf16_tn")
    return out


@gemm_tn.register_fake
def _(a, b, bias):
    return a.new_empty(a.shape[0], b.shape[0])


def _gemm_setup(ctx, inputs, output):
    a, b, bias = inputs
    ctx.save_for_backward(a, b)
    ctx.has_bias = bias is not None


def _gemm_backward(ctx, dy):
    a, b = ctx.saved_tensors
    dy = _c(dy)
    da = db = dbias = None
    if ctx.needs_input_grad[0]:
        da = gemm_tn(dy, b.t().contiguous(), None)          # dA[M,K] = dY[M,N] · (B^T)[K,N]^T
    if ctx.needs_input_grad[1]:
        db = gemm_tn(dy.t().contiguous(), a.t().contiguous(), None)   # dB[N,K] = (dY^T)[N,M] · (A^T)[K,M]^T
    if ctx.has_bias and ctx.needs_input_grad[2]:
        dbias = dy.sum(0)
    return da, db, dbias


gemm_tn.register_autograd(_gemm_backward, setup_context=_gemm_setup)


# ===================================================================================================== helpers of the linear layers
@custom_op(f"{NS}::rope_table", mutates_args=(), device_types="cuda")
def rope_table(position_ids: Tensor, inv_freq: Tensor, attention_scaling: float, dtype: torch.dtype) -> Tuple[Tensor, Tensor]:
    """int64 positions [n] x fp32 inv_freq [half] -> cos / sin [n, half] in `dtype` (once per forward)."""
    n, half = position_ids.numel(), inv_freq.numel()
    cos = torch.empty(n, half, dtype=dtype, device=position_ids.device)
    sin = torch.empty_like(cos)
    _C.check(_lib().tn_rope_table(_p(position_ids), _p(inv_freq), _p(cos), _p(sin), n, half, float(attention_scaling),
                                  _C.dcode(cos), _cur()), "tn_rope_table")
    return cos, sin


@rope_table.register_fake
def _(position_ids, inv_freq, attention_scaling, dtype):
    e = lambda: position_ids.new_empty(position_ids.numel(), inv_freq.numel(), dtype=dtype)
    return e(), e()


@custom_op(f"{NS}::transpose_bf16_", mutates_args=("dst",), device_types="cuda")
def transpose_bf16_(src: Tensor, dst: Tensor) -> None:
    """dst[c, r] = src[r, c]; both 2-D bf16 with contiguous rows (row strides are passed to the kernel)."""
    R, Cn = src.shape
    _C.check(_lib().tn_transpose_bf16(_p(src), _p(dst), R, Cn, src.stride(0), dst.stride(0), _cur()),
             "tn_transpose_bf16")


@custom_op(f"{NS}::colsum_bf16", mutates_args=(), device_types="cuda")
def colsum_bf16(x: Tensor) -> Tensor:
    """x.sum(0) of a bf16 [rows, cols] matrix (fp32 accumulation, deterministic two-stage): the bias gradient."""
    R, Cn = x.shape
    ws = torch.empty(int(_lib().tn_colsum_workspace_floats(R, Cn)), dtype=torch.float32, device=x.device)
    out = torch.empty(Cn, dtype=x.dtype, device=x.device)
    _C.check(_lib().tn_colsum_bf16(_p(x), _p(out), _p(ws), R, Cn, x.stride(0), _cur()), "tn_colsum_bf16")
    return out


@colsum_bf16.register_fake
def _(x):
    return x.new_empty(x.shape[1])


@custom_op(f"{NS}::swiglu_fwd_t", mutates_args=(), device_types="cuda")
def swiglu_fwd_t(gate: Tensor, up: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (act [M, I], act^T [I, M]): the transposed copy feeds the weight-gradient GEMM of down_proj."""
    M, I = gate.shape
    act = torch.empty_like(gate)
    act_t = torch.empty(I, M, dtype=gate.dtype, device=gate.device)
    _C.check(_lib().tn_swiglu_fwd_t(_p(gate), _p(up), _p(act), _p(act_t), M, I, _cur()), "tn_swiglu_fwd_t")
    return act, act_t


@swiglu_fwd_t.register_fake
def _(gate, up):
    return torch.empty_like(gate), gate.new_empty(gate.shape[1], gate.shape[0])


@custom_op(f"{NS}::swiglu_bwd_t", mutates_args=(), device_types="cuda")
def swiglu_bwd_t(dact: Tensor, gate: Tensor, up: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (dgate, dup, [dgate^T ; dup^T] as [2 I, M])."""
    M, I = gate.shape
    dgate, dup = torch.empty_like(gate), torch.empty_like(up)
    dgu_t = torch.empty(2 * I, M, dtype=gate.dtype, device=gate.device)
    _C.check(_lib().tn_swiglu_bwd_t(_p(dact), _p(gate), _p(up), _p(dgate), _p(dup), _p(dgu_t), M, I, _cur()),
             "tn_swiglu_bwd_t")
    return dgate, dup, dgu_t


@swiglu_bwd_t.register_fake
def _(dact, gate, up):
    return torch.empty_like(gate), torch.empty_like(up), gate.new_empty(2 * gate.shape[1], gate.shape[0])


@custom_op(f"{NS}::ce_fwd_rows", mutates_args=(), device_types="cuda")
def ce_fwd_rows(logits: Tensor, labels: Tensor, sentence_lens: Tensor, num_sentence: Tensor,
                ignore_index: int) -> Tuple[Tensor, Tensor, Tensor]:
    """Per-row part of the CE (the chunks of the fused lm_head + CE): -> (nll, lse fp32 [n], hit int32 [n])."""
    n, V = logits.shape
    nll = torch.empty(n, dtype=torch.float32, device=logits.device)
    lse = torch.empty_like(nll)
    hit = torch.empty(n, dtype=torch.int32, device=logits.device)
    _C.check(_lib().tn_ce_forward(_p(logits), _p(labels), _p(sentence_lens), _p(num_sentence), _p(nll), _p(lse),
                                  _p(hit), None, n, V, int(ignore_index), _C.dcode(logits), _cur()), "tn_ce_forward")
    return nll, lse, hit


@ce_fwd_rows.register_fake
def _(logits, labels, sentence_lens, num_sentence, ignore_index):
    n = logits.shape[0]
    return (logits.new_empty(n, dtype=torch.float32), logits.new_empty(n, dtype=torch.float32),
            logits.new_empty(n, dtype=torch.int32))


@custom_op(f"{NS}::ce_reduce", mutates_args=(), device_types="cuda")
def ce_reduce(nll: Tensor, hit: Tensor, labels: Tensor, sentence_lens: Tensor, num_sentence: Tensor,
              ignore_index: int) -> Tensor:
    """-> stats fp32 [4] = {loss_per_sample, loss_per_token, accuracy, n_valid} from the per-row parts."""
    out = torch.empty(4, dtype=torch.float32, device=nll.device)
    _C.check(_lib().tn_ce_reduce(_p(nll), _p(hit), _p(labels), _p(sentence_lens), _p(num_sentence), _p(out),
                                 nll.numel(), int(ignore_index), _cur()), "tn_ce_reduce")
    return out


@ce_reduce.register_fake
def _(nll, hit, labels, sentence_lens, num_sentence, ignore_index):
    return nll.new_empty(4)


# ===================================================================================================== audio frontend
@custom_op(f"{NS}::kaldi_fbank", mutates_args=(), device_types="cuda")
def kaldi_fbank(wav: Tensor, num_mel_bins: int) -> Tensor:
    nf = _lib().tn_fbank_frames(wav.numel())
    feat = torch.empty(nf, num_mel_bins, dtype=torch.float32, device=wav.device)
    _C.check(_lib().tn_kaldi_fbank(_p(wav), _p(feat), wav.numel(), num_mel_bins, _cur()), "tn_kaldi_fbank")
    return feat


@kaldi_fbank.register_fake
def _(wav, num_mel_bins):
    n = wav.numel()
    return wav.new_empty(max(0, 1 + (n - 400) // 160) if n >= 400 else 0, num_mel_bins)


@custom_op(f"{NS}::log_mel", mutates_args=(), device_types="cuda")
def log_mel(wav: Tensor, mel_fb: Tensor, num_mel_bins: int) -> Tensor:
    feat = torch.empty(wav.numel() // 160, num_mel_bins, dtype=torch.float32, device=wav.device)
    gmax = torch.empty(1, dtype=torch.float32, device=wav.device)
    _C.check(_lib().tn_log_mel(_p(wav), _p(mel_fb), _p(feat), _p(gmax), wav.numel(), num_mel_bins, _cur()),
             "tn_log_mel")
    return feat


@log_mel.register_fake
def _(wav, mel_fb, num_mel_bins):
    return wav.new_empty(wav.numel() // 160, num_mel_bins)


@custom_op(f"{NS}::audiofeat_stack", mutates_args=(), device_types="cuda")
def audiofeat_stack(feat: Tensor, stack: int, stride: int, normalize: bool) -> Tensor:
    T, F = feat.shape
    out = torch.empty((T + stride - 1) // stride, F * stack, dtype=torch.float32, device=feat.device)
    _C.check(_lib().tn_audiofeat_stack(_p(feat), _p(out), T, F, stack, stride, int(normalize), _cur()),
             "tn_audiofeat_stack")
    return out


@audiofeat_stack.register_fake
def _(feat, stack, stride, normalize):
    return feat.new_empty((feat.shape[0] + stride - 1) // stride, feat.shape[1] * stack)


def resample_polyphase(wave: Tensor, tab: Tensor, p: int, q: int, n_out: int) -> Tensor:
    """tn_resample_polyphase on a 1-D fp32 device waveform"""
    out = torch.empty(n_out, dtype=torch.float32, device=wave.device)
    _C.check(_lib().tn_resample_polyphase(_p(wave), _p(out), _p(tab), wave.numel(), int(n_out), int(p), int(q),
                                          int(tab.shape[1]), _cur()), "tn_resample_polyphase")
    return out


def feat_augment(feat: Tensor, t_masks, f_masks, subs, out_rows: int) -> Tensor:
    """tn_feat_augment: the draws travel as kernel arguments (host int arrays), so this is a plain function, not a
    registered op (no tensor carries them)."""
    import ctypes as C
    T, F = feat.shape
    out = torch.empty(out_rows, F, dtype=torch.float32, device=feat.device)

    def arr(rows, width):
        flat = [int(v) for r in rows for v in r]
        assert len(flat) == width * len(rows)
        return (C.c_int * max(1, len(flat)))(*flat), len(rows)
    tm, nt = arr(t_masks, 2)
    fm, nf = arr(f_masks, 2)
    sb, ns = arr(subs, 3)
    _C.check(_lib().tn_feat_augment(_p(feat), _p(out), T, F, int(out_rows), C.cast(tm, C.c_void_p), nt,
                                    C.cast(fm, C.c_void_p), nf, C.cast(sb, C.c_void_p), ns, _cur()), "tn_feat_augment")
    return out


@custom_op(f"{NS}::pcm16_to_f32", mutates_args=(), device_types="cuda")
def pcm16_to_f32(pcm: Tensor) -> Tensor:
    out = torch.empty(pcm.shape, dtype=torch.float32, device=pcm.device)
    _C.check(_lib().tn_pcm16_to_f32(_p(pcm), _p(out), pcm.numel(), _cur()), "tn_pcm16_to_f32")
    return out


@pcm16_to_f32.register_fake
def _(pcm):
    return pcm.new_empty(pcm.shape, dtype=torch.float32)


@custom_op(f"{NS}::bestrq_tokenize", mutates_args=(), device_types="cuda")
def bestrq_tokenize(feat: Tensor, quantizer: Tensor, codebook: Tensor) -> Tensor:
    T, Fdim = feat.shape
    E, V = quantizer.shape[1], codebook.shape[0]
    codes = torch.empty(T, dtype=torch.int64, device=feat.device)
    _C.check(_lib().tn_bestrq_tokenize(_p(feat), _p(quantizer), _p(codebook), _p(codes), T, Fdim, E, V, _cur()),
             "tn_bestrq_tokenize")
    return codes


@bestrq_tokenize.register_fake
def _(feat, quantizer, codebook):
    return feat.new_empty(feat.shape[0], dtype=torch.int64)


OPS = ("rmsnorm_fwd", "rmsnorm_bwd", "layernorm_fwd", "layernorm_bwd", "swiglu_fwd", "swiglu_bwd", "gelu_fwd",
       "gelu_bwd", "rope_apply", "attn_fwd", "attn_bwd", "attn_fwd_bidir", "attn_bwd_bidir", "attn_bwd_stacked", "attn_build_meta", "attn_fwd_seg", "attn_fwd_seg_chunks", "attn_merge", "attn_bwd_seg", "ce_fwd",
       "ce_bwd", "ce_bwd_", "gemm_tn", "rope_table", "transpose_bf16_", "colsum_bf16", "swiglu_fwd_t", "swiglu_bwd_t",
       "ce_fwd_rows", "ce_reduce", "kaldi_fbank", "log_mel", "audiofeat_stack", "pcm16_to_f32", "bestrq_tokenize")
