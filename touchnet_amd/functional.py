"""PyTorch-ROCm custom ops over the HIP kernels (autograd.Function wrappers around the C ABI).

Every function here runs on the device through libtouchnet_amd.so and raises if that is impossible
(CPU tensor, unsupported dtype/shape, missing library).  There is no eager fallback anywhere.

Signatures mirror the modules the reference swaps (the liger precedent at
touchnet/models/llama/__init__.py:11-15): RMSNorm, rotary embedding, SwiGLU MLP activation, the
attention interface (transformers attention-function contract, SURVEY.md §8b hook 2) and the
TrainSpec loss / acc functions (hook 3).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _C
from . import library as L

_cur = _C.stream
_p = _C.ptr


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------ norms
def rms_norm(x, weight, eps, residual=None):
    """y = RMSNorm(x (+ residual)) * weight.  With ``residual`` returns ``(y, x + residual)``: the
    residual add of the decoder layer (modeling_llama.py:306-324) is fused into the norm that follows it."""
    y, h, _ = L.rmsnorm_fwd(x, residual, weight, float(eps))
    return (y, h) if residual is not None else y


def layer_norm(x, weight, bias, eps=1e-5, residual=None):
    y, h, _, _ = L.layernorm_fwd(x, residual, weight, bias, float(eps))
    return (y, h) if residual is not None else y


# ------------------------------------------------------------------------------------ activations
def swiglu(gate, up):
    """silu(gate) * up (modeling_llama.py:174-176)."""
    return L.swiglu_fwd(gate, up)


def gelu(x):
    return L.gelu_fwd(x)


# ------------------------------------------------------------------------------------ RoPE
def rope_inv_freq(head_dim: int, theta: float, scaling: Optional[dict] = None, device=None) -> torch.Tensor:
    """fp32 [head_dim/2] inverse frequencies, default or llama3-scaled
    (the table `post_init` re-derives at touchnet/models/llama/__init__.py:23-27)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    kind = (scaling or {}).get("rope_type", (scaling or {}).get("type", "default"))
    if kind == "llama3":
        factor, lo, hi = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"]
        old = scaling["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        inv_l = torch.where(wavelen > old / lo, inv / factor, inv)
        smooth = (old / wavelen - lo) / (hi - lo)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        medium = ~(wavelen < old / hi) & ~(wavelen > old / lo)
        inv = torch.where(medium, smoothed, inv_l)
    elif kind != "default":
        raise _C.KernelError(f"unsupported rope_type {kind!r}")
    return inv.to(device) if device is not None else inv


def rope_tables(position_ids: torch.Tensor, inv_freq: torch.Tensor, dtype, attention_scaling: float = 1.0):
    """cos/sin [B*T, D/2] in ``dtype`` from packed int64 position_ids [B, T] (once per forward)."""
    pos = _c(position_ids).to(torch.int64).view(-1)
    inv = _c(inv_freq).float()
    if dtype not in _C.DTYPE_CODE:
        raise _C.KernelError(f"unsupported dtype {dtype} (float32 / bfloat16 only)")
    return L.rope_table(pos, inv, float(attention_scaling), dtype)


def apply_rope(q, k, cos, sin):
    """q [B,T,Nh,D], k [B,T,Nkv,D] (the GEMM output layout — no head transpose), tables from rope_tables."""
    return L.rope_apply(q, k, cos, sin, False)


# ------------------------------------------------------------------------------------ attention
@dataclass
class PackedMask:
    """Device-side form of the packers' document-id `attention_mask`
    (touchnet/models/llama/processing_llama.py:38-40): int32 ids + per-64-tile range metadata.
    Built once per batch and shared by all layers (the reference builds its BlockMask once per forward)."""
    doc: torch.Tensor     # int32 [B, T], 0 = pad
    meta: torch.Tensor    # int32 [5 * B * ceil(T/64)]
    B: int
    T: int


def build_packed_mask(doc_ids: torch.Tensor) -> PackedMask:
    B, T = doc_ids.shape
    doc = _c(doc_ids).to(torch.int32)
    return PackedMask(doc, L.attn_build_meta(doc), B, T)


def causal_mask(B: int, T: int, device) -> PackedMask:
    """Plain causal attention (the Qwen2-Audio training path, qwen2_audio/__init__.py:231-236) =
    one document per row."""
    return build_packed_mask(torch.ones(B, T, dtype=torch.int32, device=device))


# TN_ROPE_GRAD_IN_ATTENTION=0: the rotary embedding's backward as its own pass over dq / dk again (A/B switch; same bits)
ROPE_GRAD_IN_ATTENTION = os.environ.get("TN_ROPE_GRAD_IN_ATTENTION", "1") != "0"


class _AttentionRopeGrad(torch.autograd.Function):
    """`packed_attention` of q / k that carry a rotary embedding put there by `linear_group(rope=..)`: the backward hands
    dq / dk over as gradients of the UN-rotated projections (tn_attn_bwd_rope — the transposed rotation in the attention
    kernels' epilogues instead of a pass over dq / dk), and the projection node is told not to rotate them back again
    (`linear_group(.., rope_grad_in_attention=True)`; the pair is set up in one place, models/llama Attention.forward)."""

    @staticmethod
    def forward(ctx, q, k, v, doc, meta, scale, cos, sin):
        o, lse2 = L.attn_fwd(q, k, v, doc, meta, scale)
        ctx.save_for_backward(_c(q), _c(k), _c(v), o, lse2, doc, meta, cos, sin)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse2, doc, meta, cos, sin = ctx.saved_tensors
        B, T, Nh, D = q.shape
        Nkv = k.shape[2]
        stacked = Nh == Nkv
        out = L.attn_bwd_rope(q, k, v, o, _c(do), lse2, doc, meta, ctx.scale, cos, sin, stacked)
        if stacked:                                        # (one buffer, three slices: library.attn_bwd_stacked)
            dq, dk, dv = out.unbind(0)
        else:
            dq, dk, dv = out.split([B * T * Nh * D, B * T * Nkv * D, B * T * Nkv * D])
            dq, dk, dv = dq.view(B, T, Nh, D), dk.view(B, T, Nkv, D), dv.view(B, T, Nkv, D)
        return dq, dk, dv, None, None, None, None, None


def packed_attention(q, k, v, mask: PackedMask, scale: Optional[float] = None, rope_grad=None):
    """softmax(scale * Q K^T + doc-causal mask) V with q [B,T,Nh,D], k/v [B,T,Nkv,D] -> [B,T,Nh,D].
    ``rope_grad = (cos, sin)``: q and k are ROTATED outputs of `linear_group(.., rope=(cos, sin, ..),
    rope_grad_in_attention=True)`; their gradients leave this node already rotated back (`_AttentionRopeGrad`)."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    if tuple(q.shape[:2]) != (mask.B, mask.T):
        raise _C.KernelError(f"mask built for {(mask.B, mask.T)}, got q {tuple(q.shape[:2])}")
    if rope_grad is not None:
        cos, sin = rope_grad
        return _AttentionRopeGrad.apply(q, k, v, mask.doc, mask.meta, float(scale), _c(cos.detach()), _c(sin.detach()))
    return L.attn_fwd(q, k, v, mask.doc, mask.meta, float(scale))[0]


class _BidirectionalAttention(torch.autograd.Function):
    """softmax over ALL keys of the query's document (no causal restriction): `tn_attn_fwd_bidir` / `tn_attn_bwd_bidir`,
    the packed-attention kernels with the causal term of the predicate switched off and the key range of a query tile
    extended to the last tile that shares a document with it (csrc/attn_common.h QView::bidir).
    Used by the Whisper speech encoder of Kimi-Audio (touchnet/models/kimi_audio/modeling_kimi_audio.py:337-339, 942-947:
    transformers' WhisperEncoder, bidirectional self-attention over the 1500 frames of a clip).
    (Round 4 first built it from two launches of the causal kernels — the batch and its time reversal — merged by their
    log-sum-exp in torch: correct, but the flips and the fp32 merge were 32 % of the speech workload's step.)"""

    @staticmethod
    def forward(ctx, q, k, v, doc, meta, scale):
        q, k, v = _c(q), _c(k), _c(v)
        o, lse = L.attn_fwd_bidir(q, k, v, doc, meta, scale)
        ctx.save_for_backward(q, k, v, o, lse, doc, meta)
        ctx.scale = scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, doc, meta = ctx.saved_tensors
        dq, dk, dv = L.attn_bwd_bidir(q, k, v, o, _c(do), lse, doc, meta, ctx.scale)
        return dq, dk, dv, None, None, None


def bidirectional_attention(q, k, v, mask: PackedMask, scale: Optional[float] = None):
    """softmax(scale * Q K^T + same-document mask) V — every key of the query's document, before AND after it — with
    q [B,T,Nh,D], k/v [B,T,Nkv,D] -> [B,T,Nh,D]; pad rows (document 0) give 0."""
    if scale is None:
        scale = q.shape[-1] ** -0.5
    if tuple(q.shape[:2]) != (mask.B, mask.T):
        raise _C.KernelError(f"mask built for {(mask.B, mask.T)}, got q {tuple(q.shape[:2])}")
    return _BidirectionalAttention.apply(q, k, v, mask.doc, mask.meta, float(scale))


@dataclass(frozen=True)
class SeqShard:
    """Which global positions of the packed row the local query-side rows hold (context parallelism).
    `segs` = up to two (row0, rows, global_offset) triples; buffers are [B, rows_per_batch, ...]."""
    segs: tuple
    rows_per_batch: int

    def flat(self):
        return [int(v) for seg in self.segs for v in seg]


def packed_attention_sharded(q_local, k_full, v_full, mask: PackedMask, shard: SeqShard,
                             scale: Optional[float] = None):
    """Context-parallel building block: local query rows (per `shard`) against the full, all-gathered K/V.
    The gradient w.r.t. k_full / v_full is this rank's PARTIAL sum (to be reduce-scattered by the caller)."""
    if scale is None:
        scale = q_local.shape[-1] ** -0.5
    return L.attn_fwd_seg(q_local, k_full, v_full, mask.doc, mask.meta, float(scale), shard.flat(),
                          int(shard.rows_per_batch))[0]


class _SplitAttention(torch.autograd.Function):
    """Context-parallel attention with the exchange hidden under the LOCAL part: the rank's query rows first attend to
    the keys of its OWN chunks (in the global K/V buffers from the start), then the compute stream waits for the halo
    (`wait`), the rows attend to the RECEIVED chunks, and the two partial results are merged by their log-sum-exp
    (tn_attn_merge).  The backward is ONE tn_attn_bwd_seg over all chunks with the merged O / LSE — the gradient of a
    softmax over the union of the key sets does not care how the forward was split."""

    @staticmethod
    def forward(ctx, q, k_full, v_full, doc, meta, scale, segs, rpb, chunk_len, own_mask, remote_mask, wait):
        o, lse = L.attn_fwd_seg_chunks(q, k_full, v_full, doc, meta, scale, segs, rpb, chunk_len, own_mask)
        if wait is not None:
            wait()
        if remote_mask:
            o_b, lse_b = L.attn_fwd_seg_chunks(q, k_full, v_full, doc, meta, scale, segs, rpb, chunk_len, remote_mask)
            o, lse = L.attn_merge(o, lse, o_b, lse_b)
        ctx.save_for_backward(_c(q), _c(k_full), _c(v_full), o, lse, doc, meta)
        ctx.scale, ctx.segs, ctx.rpb = scale, list(segs), rpb
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, doc, meta = ctx.saved_tensors
        dq, dk, dv = L.attn_bwd_seg(q, k, v, o, do, lse, doc, meta, ctx.scale, ctx.segs, ctx.rpb)
        return dq, dk, dv, None, None, None, None, None, None, None, None, None


def packed_attention_sharded_split(q_local, k_full, v_full, mask: PackedMask, shard: SeqShard, chunk_len: int,
                                   own_chunks, remote_chunks, wait=None, scale: Optional[float] = None):
    """`packed_attention_sharded` as two key-side parts + an LSE merge: `own_chunks` / `remote_chunks` = indices of the
    sequence chunks (of `chunk_len` positions) the first / second part covers; `wait()` is called between the two (the
    halo exchange's completion: the remote chunks of k_full / v_full may still be in flight before it)."""
    if scale is None:
        scale = q_local.shape[-1] ** -0.5
    bits = lambda cs: sum(1 << int(c) for c in cs)
    return _SplitAttention.apply(q_local, k_full, v_full, mask.doc, mask.meta, float(scale), shard.flat(),
                                 int(shard.rows_per_batch), int(chunk_len), bits(own_chunks), bits(remote_chunks), wait)


# ------------------------------------------------------------------------------------ loss
def _num_sentence_dev(num_sentence, device):
    if isinstance(num_sentence, torch.Tensor):
        return num_sentence.to(device=device, dtype=torch.float32).reshape(1)
    return torch.tensor([float(num_sentence)], dtype=torch.float32, device=device)


class _PackedCEInplace(torch.autograd.Function):
    """`inplace_grad=True`: the backward writes dlogits over the logits (dead after the loss): mi355_touch::ce_bwd_."""

    @staticmethod
    def forward(ctx, logits, lab, sl, ns, ignore_index):
        with torch.no_grad():
            loss, out, lse = L.ce_fwd(logits, lab, sl, ns, ignore_index)
        ctx.save_for_backward(logits, lab, sl, lse, ns)
        ctx.ignore_index = ignore_index
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, g_loss, _g_out):
        logits, lab, sl, lse, ns = ctx.saved_tensors
        L.ce_bwd_(logits, lab, sl, lse, ns, _c(g_loss).to(torch.float32).reshape(1), ctx.ignore_index)
        return logits, None, None, None, None


def packed_cross_entropy(pred, labels, sentence_lens, num_sentence, ignore_index=-100, inplace_grad=False):
    """Returns ``(loss_per_sample [differentiable], stats)`` with
    ``stats = [loss_per_sample, loss_per_token, accuracy, n_valid]`` (fp32 device tensor, no host sync)."""
    V = pred.shape[-1]
    logits = _c(pred).view(-1, V)
    lab = _c(labels).to(torch.int64).view(-1)
    sl = _c(sentence_lens).to(torch.int64).view(-1)
    ns = _num_sentence_dev(num_sentence, pred.device)
    if inplace_grad:
        return _PackedCEInplace.apply(logits, lab, sl, ns, int(ignore_index))
    loss, stats, _ = L.ce_fwd(logits, lab, sl, ns, int(ignore_index))
    return loss, stats.detach()


class _FusedLinearCE(torch.autograd.Function):
    """lm_head GEMM + packed CE, chunked over tokens so the [B*T, V] logits never exist at once
    (the liger fused-linear-CE idea, touchnet/bin/train.py:443-445 — but keeping the reference's
    per-sentence normalisation, which liger's mean-over-tokens drops, SURVEY.md §2.3 K10').
    Gradients w.r.t. hidden and weight are produced in the forward pass and scaled by the upstream
    gradient in backward."""

    @staticmethod
    def forward(ctx, hidden, weight, labels, sentence_lens, num_sentence, ignore_index, chunk, compact=False, tp=None):
        """`tp = (group, rank, size)` — loss parallel (models/tensor_parallel.py): `weight` holds rows
        [rank V/size, (rank+1) V/size) of the vocabulary.  Every chunk computes the LOCAL logits and row statistics with the
        unchanged CE kernels (a label outside the shard is passed as V_local = "no local target": the forward kernel then
        returns nll = lse_local, the backward kernel a pure softmax), combines max / sum-exp / target logit over the group
        (three [rows] all-reduces) and back-propagates with the GLOBAL log-sum-exp.  d(hidden) is this rank's PARTIAL sum
        over its vocabulary shard: the sequence gather in front of the head (`SequenceParallel.gather`, partial=True)
        reduce-scatters it."""
        H = hidden.shape[-1]
        h2 = hidden.reshape(-1, H)
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        sl = sentence_lens.reshape(-1).to(torch.int64).contiguous()
        rows = None
        overflow = None
        if compact is not False and compact is not None:
            # Positions whose label is ignore_index contribute exactly 0 to the loss, to d(hidden) and to d(weight) —
            # the CE kernels already never read their logits; here the lm_head GEMMs skip them too (ASR-SFT batches:
            # > 90 % of the positions are audio / prompt).  Same loss and gradients up to GEMM summation order.
            #   compact = int   upper bound of the labelled positions, known to the data loader (the packers emit
            #                   `labelled_rows_max`): static shapes, NO host synchronisation — rows beyond the real count
            #                   are given the label ignore_index; a bound that is too small poisons the loss with NaN
            #                   (decided on the device) instead of silently dropping labels
            #   compact = True  exact count read back from the device (one host sync per step)
            n_all = h2.shape[0]
            labelled = lab != ignore_index
            if compact is True:
                rows = torch.nonzero(labelled).squeeze(1)
            else:
                n_max = min(int(compact), n_all)
                rows = torch.nonzero_static(labelled, size=max(n_max, 1), fill_value=0).squeeze(1)
                count = labelled.sum()
                valid = torch.arange(rows.numel(), device=lab.device) < count
                overflow = count > n_max
            if rows.numel() == 0:                              # nothing labelled: zero loss, zero gradients
                rows = None
            else:
                h2, lab, sl = h2.index_select(0, rows), lab.index_select(0, rows), sl.index_select(0, rows)
                if compact is not True:
                    lab = torch.where(valid, lab, torch.full_like(lab, ignore_index))
        n, V = h2.shape[0], weight.shape[0]
        ns = _num_sentence_dev(num_sentence, hidden.device)
        one = torch.ones(1, dtype=torch.float32, device=hidden.device)
        dh = torch.empty_like(h2)
        dw = None
        parts = []
        lab_k = lab                                        # the labels the kernels see
        if tp is not None:
            from touchnet_amd.models.tensor_parallel import tp_all_reduce
            group, tp_rank, _ = tp
            v0 = tp_rank * V
            mine = (lab >= v0) & (lab < v0 + V)
            lab_k = torch.where(lab == ignore_index, lab, torch.where(mine, lab - v0, torch.full_like(lab, V)))
        for s in range(0, n, chunk):
            e = min(s + chunk, n)
            logits = _mm_tn(h2[s:e], weight) if (h2.is_cuda and h2.dtype == torch.bfloat16 and _bf16_rows(h2[s:e], weight)) \
                else torch.nn.functional.linear(h2[s:e], weight)             # [c, V]: the only logits alive
            nll_c, lse_c, hit_c = L.ce_fwd_rows(logits, lab_k[s:e], sl[s:e], ns, int(ignore_index))
            if tp is not None:
                valid = lab[s:e] != ignore_index
                target = lse_c - nll_c                                        # the target's logit where it is local, else 0
                top_v, top_i = logits.float().max(dim=1)
                mx = torch.where(valid, lse_c, torch.full_like(lse_c, float("-inf")))
                tp_all_reduce(mx, group, torch.distributed.ReduceOp.MAX)
                se = torch.where(valid, torch.exp(lse_c - mx), torch.zeros_like(lse_c))
                tp_all_reduce(se, group)
                tp_all_reduce(target, group)
                lse_c = torch.where(valid, mx + torch.log(se), torch.zeros_like(lse_c))
                nll_c = torch.where(valid, lse_c - target, torch.zeros_like(lse_c))
                # accuracy: the argmax over ALL shards (first index on ties: lowest shard, then lowest index inside it)
                best = torch.stack([top_v, (top_i + v0).float()], dim=1)                      # indices < 2^24: exact
                if not getattr(group, "emulated", False):
                    size = torch.distributed.get_world_size(group)
                    every = torch.empty((size * best.shape[0], 2), dtype=best.dtype, device=best.device)
                    torch.distributed.all_gather_into_tensor(every, best.contiguous(), group=group)
                    every = every.view(size, best.shape[0], 2)                                # rank-major
                    win = every[..., 0].argmax(dim=0)
                    best = every.gather(0, win[None, :, None].expand(1, -1, 2))[0]
                hit_c = ((best[:, 1].to(torch.int64) == lab[s:e]) & valid).to(hit_c.dtype)
            L.ce_bwd_(logits, lab_k[s:e], sl[s:e], lse_c, ns, one, int(ignore_index))     # logits := dlogits
            parts.append((nll_c, hit_c))
            hc = h2[s:e]
            own_d = (logits.is_cuda and _own(e - s, H, (V,)) and _bf16_rows(logits, weight, dh[s:e]))
            if own_d:
                gemm([(logits, weight)], b_kmaj=True, out=dh[s:e])            # dh = dlogits @ W (W read contraction-major)
            else:
                torch.mm(logits, weight, out=dh[s:e])
            own_w = (logits.is_cuda and _own(V, H, (e - s,), True, True) and _bf16_rows(logits, hc))
            if dw is None:
                dw = gemm([(logits, hc)], True, True) if own_w else torch.mm(logits.t(), hc)     # dW = dlogits^T @ h
            elif own_w:
                gemm([(logits, hc)], True, True, out=dw, accumulate=True)     # accumulate inside the GEMM epilogue
            else:
                dw.addmm_(logits.t(), hc)
            del logits
        nll = torch.cat([a for a, _ in parts]) if len(parts) > 1 else parts[0][0]
        hit = torch.cat([b for _, b in parts]) if len(parts) > 1 else parts[0][1]
        out = L.ce_reduce(nll, hit, lab, sl, ns, int(ignore_index))
        if rows is not None:                                   # scatter d(hidden) back; ignored rows stay 0
            # (index_add: the filler rows of the static form repeat index 0 and carry exact zeros)
            dh = torch.zeros(n_all, H, dtype=dh.dtype, device=dh.device).index_add_(0, rows, dh)
        if overflow is not None:
            # a bound that is too small poisons the statistics AND both gradients (they were computed from the truncated
            # label set): the optimizer's non-finite check then skips the step instead of applying it
            poison = torch.where(overflow, float("nan"), 1.0)
            out = out * poison.to(out.dtype)
            dh = dh * poison.to(dh.dtype)
            dw = dw * poison.to(dw.dtype) if dw is not None else dw
        ctx.save_for_backward(dh, dw)
        ctx.hshape, ctx.wdtype = hidden.shape, weight.dtype
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g):
        dh, dw = ctx.saved_tensors
        g = g_loss.to(torch.float32)
        # (same-dtype operands: a bf16 tensor times an fp32 0-dim DEVICE tensor takes TensorIterator's casting kernel,
        #  1.4 TB/s on the [V, H] weight gradient — 1.8 ms/step at V = 156 k; the upstream gradient is 1 or a power of two)
        return ((dh * g.to(dh.dtype)).view(ctx.hshape), (dw * g.to(dw.dtype)).to(ctx.wdtype), None, None, None, None, None,
                None, None)


def fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence, ignore_index=-100,
                               chunk_tokens=4096, compact=False, tp=None):
    """Returns ``(loss_per_sample [differentiable], stats)`` like packed_cross_entropy, from hidden states.
    ``compact``: run lm_head only on the labelled positions — an int upper bound from the data loader (no host sync)
    or True (exact, one sync); ``tp``: vocabulary-parallel head (loss parallel); see _FusedLinearCE.forward."""
    return _FusedLinearCE.apply(hidden, weight, labels, sentence_lens, num_sentence, ignore_index, chunk_tokens,
                                compact, tp)


# ------------------------------------------------------------------------------------ linear layers
def transpose_2d(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[c, r] = x[r, c]`` for a bf16 matrix whose rows are contiguous (row stride >= cols), HIP kernel."""
    if x.dim() != 2 or x.dtype != torch.bfloat16 or not (x.is_cuda or x.is_meta) or x.stride(1) != 1:
        raise RuntimeError("transpose_2d: expects a 2-D bf16 device tensor with contiguous rows")
    R, Cn = x.shape
    if out is None:
        out = torch.empty(Cn, R, dtype=x.dtype, device=x.device)
    L.transpose_bf16_(x, out)
    return out


def column_sum(x: torch.Tensor) -> torch.Tensor:
    """``x.sum(0)`` of a bf16 [rows, cols] device matrix (fp32 accumulation, bf16 result): the bias gradient."""
    R, Cn = x.shape
    if x.dtype != torch.bfloat16 or not (x.is_cuda or x.is_meta) or x.stride(1) != 1 or Cn % 8 or x.stride(0) % 8:
        return x.sum(0)                                    # (fp32 / odd widths: torch's reduction, still on the device)
    return L.colsum_bf16(x)


def gemm_tn(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            accumulate: bool = False, out_t: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out = a @ b.T (+ bias) (+ out)`` with the hand-written MFMA kernel (csrc/gemm.hip): a [M, K], b [N, K] bf16
    device matrices with contiguous rows.  ``out_t`` [N, M] additionally receives the transposed result."""
    if a.dim() != 2 or b.dim() != 2 or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
        raise _C.KernelError("gemm_tn: 2-D bf16 operands only")
    if a.stride(1) != 1 or b.stride(1) != 1 or a.shape[1] != b.shape[1]:
        raise _C.KernelError("gemm_tn: operands must be contraction-contiguous [M, K] / [N, K]")
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        if accumulate:
            raise _C.KernelError("gemm_tn: accumulate needs `out`")
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise _C.KernelError("gemm_tn: bad `out`")
    if bias is not None:
        bias = _c(bias).to(a.dtype)
    _C.check(_C.lib().tn_gemm_bf16_tn(_p(a), _p(b), _p(out), _p(out_t), _p(bias), M, N, K, a.stride(0), b.stride(0),
                                      out.stride(0), out_t.stride(0) if out_t is not None else 0, int(accumulate),
                                      _cur()), "tn_gemm_bf16_tn")
    return out


def gemm(segs, a_kmaj: bool = False, b_kmaj: bool = False, bias: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, accumulate: bool = False,
         out_t: Optional[torch.Tensor] = None, bias_grad: Optional[torch.Tensor] = None,
         addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out[M, N] = sum_s opA_s @ opB_s^T (+ bias) (+ out)`` on the hand-written MFMA kernel (csrc/gemm.hip), fp32
    accumulation over all segments, one rounding.  ``segs`` = 1..3 pairs ``(a, b)`` of bf16 device matrices with
    contiguous rows AS STORED: ``a`` is [M, K] (``a_kmaj=False``) or [K, M] (``a_kmaj=True``: the contraction index is the
    slow one), ``b`` is [N, K] or [K, N] likewise.  The three products of a linear layer y = x W^T:
        forward          gemm([(x, W)])                          x [M, K], W [N, K]
        input gradient   gemm([(dy, W)], b_kmaj=True)            dy [M, N] · W [N, K]      (contraction over N)
        weight gradient  gemm([(dy, x)], True, True)             dy [M, N]^T · x [M, K]    (contraction over tokens)
    — none of them needs a transposed copy of an operand.
    ``bias_grad`` (weight-gradient mode, one segment): a bf16 [M] vector that receives the column sums of ``a`` — the bias
    gradient dY.sum(0) — from the same launch (tn_gemm_bf16_wgrad_bias): no separate pass over dY.
    ``addend`` (one segment, bf16 [M, N], contiguous rows): ``out = bf16(product + bias) + addend`` in the epilogue
    (tn_gemm_bf16_addend: the residual stream added where the projection is produced)."""
    if not 1 <= len(segs) <= 3:
        raise _C.KernelError("gemm: 1..3 segments")
    for a, b in segs:
        if a.dim() != 2 or b.dim() != 2 or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
            raise _C.KernelError("gemm: 2-D bf16 operands only")
        if a.stride(1) != 1 or b.stride(1) != 1:
            raise _C.KernelError("gemm: operands need contiguous rows")
    a0, b0 = segs[0]
    M = a0.shape[1] if a_kmaj else a0.shape[0]
    N = b0.shape[1] if b_kmaj else b0.shape[0]
    Ks = []
    for a, b in segs:
        Ka, Ma = (a.shape[0], a.shape[1]) if a_kmaj else (a.shape[1], a.shape[0])
        Kb, Nb = (b.shape[0], b.shape[1]) if b_kmaj else (b.shape[1], b.shape[0])
        if Ka != Kb or Ma != M or Nb != N:
            raise _C.KernelError(f"gemm: segment shapes disagree: a {tuple(a.shape)} b {tuple(b.shape)}")
        Ks.append(Ka)
    if out is None:
        if accumulate:
            raise _C.KernelError("gemm: accumulate needs `out`")
        out = torch.empty(M, N, dtype=torch.bfloat16, device=a0.device)
    if out.dtype == torch.float32:
        # weight gradient straight into an fp32 buffer (the data-parallel engine's reduce-scatter input)
        if not (a_kmaj and b_kmaj and len(segs) == 1 and bias is None and out_t is None):
            raise _C.KernelError("gemm: fp32 output exists for the single-segment weight-gradient mode only")
        if out.stride(1) != 1 or tuple(out.shape) != (M, N) or not a0.is_cuda:
            raise _C.KernelError("gemm: bad fp32 `out`")
        split = split_k(M, N, Ks[0], True, True) if SPLIT_K else 1
        ws = None
        if split > 1:
            ws = torch.empty(split * (((M + 255) // 256) * ((N + 255) // 256) * 65536 + (M + 255) // 256 * 256),
                             dtype=torch.float32, device=a0.device)
        if bias_grad is not None:
            _check_bias_grad(bias_grad, M, a0)
            _C.check(_C.lib().tn_gemm_bf16_wgrad_bias(_p(a0), _p(b0), a0.stride(0), b0.stride(0), Ks[0], _p(out), _p(bias_grad),
                                                      M, N, out.stride(0), int(accumulate), 1, split, _p(ws),
                                                      ws.numel() * 4 if ws is not None else 0, _cur()),
                     "tn_gemm_bf16_wgrad_bias")
            return out
        _C.check(_C.lib().tn_gemm_bf16_wgrad_f32(_p(a0), _p(b0), a0.stride(0), b0.stride(0), Ks[0], _p(out), M, N,
                                                 out.stride(0), int(accumulate), split, _p(ws),
                                                 ws.numel() * 4 if ws is not None else 0, _cur()), "tn_gemm_bf16_wgrad_f32")
        return out
    if out.stride(1) != 1 or tuple(out.shape) != (M, N) or out.dtype != torch.bfloat16:
        raise _C.KernelError("gemm: bad `out`")
    if bias is not None:
        bias = _c(bias).to(torch.bfloat16)
    n = len(segs)
    if not a0.is_cuda:
        raise _C.KernelError("touchnet_amd kernels need device (HIP) tensors; got a CPU tensor")
    if addend is not None:
        if (n != 1 or accumulate or out_t is not None or bias_grad is not None or addend.dtype != torch.bfloat16
                or tuple(addend.shape) != (M, N) or addend.stride(1) != 1 or addend.data_ptr() == out.data_ptr()):
            raise _C.KernelError("gemm: addend needs one segment, a bf16 [M, N] matrix with contiguous rows that is not `out`")
        _C.check(_C.lib().tn_gemm_bf16_addend(_p(a0), _p(b0), a0.stride(0), b0.stride(0), Ks[0], int(a_kmaj), int(b_kmaj),
                                              _p(out), _p(bias), _p(addend), addend.stride(0), M, N, out.stride(0), _cur()),
                 "tn_gemm_bf16_addend")
        return out
    split = 1
    if bias_grad is not None:
        if not (a_kmaj and b_kmaj and n == 1 and bias is None and out_t is None):
            raise _C.KernelError("gemm: bias_grad exists for the single-segment weight-gradient mode only")
        _check_bias_grad(bias_grad, M, a0)
        split = split_k(M, N, Ks[0], True, True) if SPLIT_K else 1
        ws = None
        if split > 1:
            ws = torch.empty(split * (((M + 255) // 256) * ((N + 255) // 256) * 65536 + (M + 255) // 256 * 256),
                             dtype=torch.float32, device=a0.device)
        _C.check(_C.lib().tn_gemm_bf16_wgrad_bias(_p(a0), _p(b0), a0.stride(0), b0.stride(0), Ks[0], _p(out), _p(bias_grad), M,
                                                  N, out.stride(0), int(accumulate), 0, split, _p(ws),
                                                  ws.numel() * 4 if ws is not None else 0, _cur()), "tn_gemm_bf16_wgrad_bias")
        return out
    if n == 1 and out_t is None and SPLIT_K:
        split = split_k(M, N, Ks[0], a_kmaj, b_kmaj)
    if split > 1:
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        ws = torch.empty(split * tiles * 65536, dtype=torch.float32, device=a0.device)
        _C.check(_C.lib().tn_gemm_bf16_splitk(_p(a0), _p(b0), a0.stride(0), b0.stride(0), Ks[0], int(a_kmaj), int(b_kmaj),
                                              _p(out), _p(bias), M, N, out.stride(0), int(accumulate), split, 0,
                                              _p(ws), ws.numel() * 4, _cur()), "tn_gemm_bf16_splitk")
        return out
    import ctypes as C
    Ap = (C.c_void_p * n)(*[a.data_ptr() for a, _ in segs])
    Bp = (C.c_void_p * n)(*[b.data_ptr() for _, b in segs])
    la = (C.c_longlong * n)(*[a.stride(0) for a, _ in segs])
    lb = (C.c_longlong * n)(*[b.stride(0) for _, b in segs])
    Kc = (C.c_int * n)(*Ks)
    _C.check(_C.lib().tn_gemm_bf16(Ap, Bp, la, lb, Kc, n, int(a_kmaj), int(b_kmaj), _p(out), _p(out_t), _p(bias), M, N,
                                   out.stride(0), out_t.stride(0) if out_t is not None else 0, int(accumulate),
                                   _cur()), "tn_gemm_bf16")
    return out


def _check_bias_grad(bias_grad, M, like):
    if (bias_grad.dtype != torch.bfloat16 or bias_grad.dim() != 1 or bias_grad.numel() != M or not bias_grad.is_contiguous()
            or bias_grad.device != like.device):
        raise _C.KernelError("gemm: bias_grad must be a contiguous bf16 [M] device vector")


def gemm_grouped_wgrad(pairs, outs=None, accumulate: bool = False):
    """``out_g[M_g, N_g] (+)= a_g^T @ b_g`` for 1..3 independent pairs of bf16 device matrices stored [K_g, M_g] / [K_g, N_g]
    (contraction-major: the weight gradients dY^T x of linear layers) as ONE persistent launch of the hand-written kernel
    (tn_gemm_bf16_grouped): the tile lists of the products are concatenated, so that only the remainder of the WHOLE list
    — not of every product — is left for a partial last round, and that remainder runs split-K.  ``outs``: bf16 or fp32
    [M_g, N_g] matrices (all of one dtype) to write (or, ``accumulate``, to add to); default new bf16 tensors."""
    import ctypes as C
    n = len(pairs)
    if not 1 <= n <= 3:
        raise _C.KernelError("gemm_grouped_wgrad: 1..3 products")
    for a, b in pairs:
        if (a.dim() != 2 or b.dim() != 2 or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.stride(1) != 1
                or b.stride(1) != 1 or a.shape[0] != b.shape[0] or not a.is_cuda):
            raise _C.KernelError("gemm_grouped_wgrad: pairs of 2-D bf16 device matrices [K, M] / [K, N] with contiguous rows")
    if outs is None:
        if accumulate:
            raise _C.KernelError("gemm_grouped_wgrad: accumulate needs `outs`")
        outs = [torch.empty(a.shape[1], b.shape[1], dtype=torch.bfloat16, device=a.device) for a, b in pairs]
    f32 = outs[0].dtype == torch.float32
    for o, (a, b) in zip(outs, pairs):
        if (o.dtype != outs[0].dtype or o.dtype not in (torch.float32, torch.bfloat16) or o.stride(1) != 1
                or tuple(o.shape) != (a.shape[1], b.shape[1])):
            raise _C.KernelError("gemm_grouped_wgrad: bad `outs`")
    Ms = (C.c_int * n)(*[a.shape[1] for a, _ in pairs])
    Ns = (C.c_int * n)(*[b.shape[1] for _, b in pairs])
    Ks = (C.c_int * n)(*[a.shape[0] for a, _ in pairs])
    need = int(_C.lib().tn_gemm_grouped_workspace_bytes(Ms, Ns, Ks, n))
    ws = torch.empty(need // 4, dtype=torch.float32, device=outs[0].device) if need > 0 else None
    Ap = (C.c_void_p * n)(*[a.data_ptr() for a, _ in pairs])
    Bp = (C.c_void_p * n)(*[b.data_ptr() for _, b in pairs])
    Cp = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    la = (C.c_longlong * n)(*[a.stride(0) for a, _ in pairs])
    lb = (C.c_longlong * n)(*[b.stride(0) for _, b in pairs])
    lc = (C.c_longlong * n)(*[o.stride(0) for o in outs])
    _C.check(_C.lib().tn_gemm_bf16_grouped(Ap, Bp, la, lb, Ks, Cp, lc, Ms, Ns, n, 1, 1, int(accumulate), int(f32), _p(ws),
                                           need, _cur()), "tn_gemm_bf16_grouped")
    return list(outs)


def gemm_swiglu_fwd(x2: torch.Tensor, w_gate: torch.Tensor, w_up: torch.Tensor):
    """``(gate, up, act)`` with gate = x2 @ w_gate^T, up = x2 @ w_up^T, act = silu(gate) * up from ONE launch of the
    hand-written kernel whose epilogue applies the SwiGLU (tn_gemm_bf16_swiglu_fwd; bit-identical to the two products
    followed by `swiglu`)."""
    M, K = x2.shape
    I = w_gate.shape[0]
    if not _bf16_rows(x2, w_gate, w_up) or w_up.shape != w_gate.shape or w_gate.shape[1] != K or w_gate.stride(0) != w_up.stride(0):
        raise _C.KernelError("gemm_swiglu_fwd: bf16 device matrices x [M, K], w_gate / w_up [I, K] of one row pitch")
    gate, up, act = (torch.empty(M, I, dtype=torch.bfloat16, device=x2.device) for _ in range(3))
    _C.check(_C.lib().tn_gemm_bf16_swiglu_fwd(_p(x2), _p(w_gate), _p(w_up), _p(gate), _p(up), _p(act), M, I, K, x2.stride(0),
                                              w_gate.stride(0), I, _cur()), "tn_gemm_bf16_swiglu_fwd")
    return gate, up, act


def gemm_gelu_fwd(x2: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]):
    """``(pre, act)`` with pre = x2 @ w^T + bias and act = gelu(pre) (exact erf) from ONE launch (tn_gemm_bf16_gelu_fwd;
    bit-identical to the product followed by `gelu`)."""
    M, K = x2.shape
    N = w.shape[0]
    if not _bf16_rows(x2, w) or w.shape[1] != K or (bias is not None and (bias.dtype != torch.bfloat16 or bias.numel() != N)):
        raise _C.KernelError("gemm_gelu_fwd: bf16 device matrices x [M, K], w [N, K], bias [N]")
    pre, act = (torch.empty(M, N, dtype=torch.bfloat16, device=x2.device) for _ in range(2))
    _C.check(_C.lib().tn_gemm_bf16_gelu_fwd(_p(x2), _p(w), _p(_c(bias)) if bias is not None else None, _p(pre), _p(act), M, N,
                                            K, x2.stride(0), w.stride(0), N, _cur()), "tn_gemm_bf16_gelu_fwd")
    return pre, act


def gemm_gelu_bwd(dy2: torch.Tensor, w2: torch.Tensor, pre: torch.Tensor):
    """``d(pre) = (dy2 @ w2) o gelu'(pre)`` (w2 [H, I] read contraction-major): one launch, d(act) never reaches HBM
    (tn_gemm_bf16_gelu_bwd; bit-identical to the product followed by `gelu_bwd`)."""
    M, H = dy2.shape
    I = w2.shape[1]
    if not _bf16_rows(dy2, w2, pre) or w2.shape[0] != H or tuple(pre.shape) != (M, I) or pre.stride(0) != I:
        raise _C.KernelError("gemm_gelu_bwd: bf16 device matrices dy [M, H], w2 [H, I], dense pre [M, I]")
    dpre = torch.empty(M, I, dtype=torch.bfloat16, device=dy2.device)
    _C.check(_C.lib().tn_gemm_bf16_gelu_bwd(_p(dy2), _p(w2), _p(pre), _p(dpre), M, I, H, dy2.stride(0), w2.stride(0), I,
                                            _cur()), "tn_gemm_bf16_gelu_bwd")
    return dpre


def gemm_swiglu_bwd(dy2: torch.Tensor, w_down: torch.Tensor, gate: torch.Tensor, up: torch.Tensor):
    """``(d_gate, d_up)`` of act = silu(gate) * up for d(act) = dy2 @ w_down (w_down [H, I] read contraction-major): one
    launch, d(act) never reaches HBM (tn_gemm_bf16_swiglu_bwd; bit-identical to the product followed by `swiglu_bwd`)."""
    M, H = dy2.shape
    I = w_down.shape[1]
    if (not _bf16_rows(dy2, w_down, gate, up) or w_down.shape[0] != H or tuple(gate.shape) != (M, I) or gate.shape != up.shape
            or gate.stride(0) != up.stride(0)):
        raise _C.KernelError("gemm_swiglu_bwd: bf16 device matrices dy [M, H], w_down [H, I], gate / up [M, I]")
    dgate = torch.empty(M, I, dtype=torch.bfloat16, device=dy2.device)
    dup = torch.empty(M, I, dtype=torch.bfloat16, device=dy2.device)
    if gate.stride(0) != I:
        raise _C.KernelError("gemm_swiglu_bwd: gate / up must be dense [M, I]")
    _C.check(_C.lib().tn_gemm_bf16_swiglu_bwd(_p(dy2), _p(w_down), _p(gate), _p(up), _p(dgate), _p(dup), M, I, H,
                                              dy2.stride(0), w_down.stride(0), I, _cur()), "tn_gemm_bf16_swiglu_bwd")
    return dgate, dup


def gemm_rope(x2: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], cos: torch.Tensor, sin: torch.Tensor,
              head_dim: int) -> torch.Tensor:
    """``rope(x2 @ weight^T + bias)`` [M, heads * head_dim] from ONE launch (tn_gemm_bf16_rope): the rotary embedding of a
    q / k projection in the GEMM's epilogue, bit-identical to the projection followed by `apply_rope`.  cos / sin: the
    bf16 [M, head_dim / 2] tables of `rope_tables`."""
    M, K = x2.shape
    N = weight.shape[0]
    if (not _bf16_rows(x2, weight) or weight.shape[1] != K or cos.dtype != torch.bfloat16 or sin.dtype != torch.bfloat16
            or tuple(cos.shape) != (M, head_dim // 2) or tuple(sin.shape) != (M, head_dim // 2)
            or not cos.is_contiguous() or not sin.is_contiguous()):
        raise _C.KernelError("gemm_rope: bf16 x [M, K], weight [N, K], cos / sin [M, head_dim / 2]")
    if bias is not None:
        bias = _c(bias).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=x2.device)
    _C.check(_C.lib().tn_gemm_bf16_rope(_p(x2), _p(weight), _p(bias), _p(cos), _p(sin), _p(out), M, N, K, x2.stride(0),
                                        weight.stride(0), N, int(head_dim), _cur()), "tn_gemm_bf16_rope")
    return out


def rope_epilogue_ok(M: int, N: int, K: int, head_dim: int) -> bool:
    """shapes tn_gemm_bf16_rope takes (and the hand-written kernel is the configured GEMM for)"""
    return (ROPE_EPILOGUE and _own(M, N, (K,)) and K % 64 == 0
            and (not SPLIT_K or split_k(M, N, K, False, False) == 1)     # (a split-K product sums in another order)
            and ((head_dim == 128 and N % 256 == 0) or (head_dim == 64 and N % 64 == 0))
            and not os.environ.get("TN_GEMM_VARIANT"))


ROPE_EPILOGUE = os.environ.get("TN_ROPE_EPILOGUE", "1") != "0"          # (A/B switch)

# A/B switches of the round-5 fusions (measurements; both default on): SwiGLU inside the gate/up and down-dgrad epilogues,
# the MLP's three weight gradients as one grouped launch
MLP_EPILOGUE = os.environ.get("TN_MLP_EPILOGUE", "1") != "0"
GROUPED_WGRAD = os.environ.get("TN_GROUPED_WGRAD", "1") != "0"

SPLIT_K = os.environ.get("TN_GEMM_SPLITK", "1") != "0"      # (A/B switch)
# (Splitting only the LAST PARTIAL ROUND of a many-tile product — tn_gemm_bf16_splitk's tail_only = 1 — was measured
# neutral-to-negative on the step in round 5 and its switch is gone; the grouped weight gradient's remainder split in
# tn_gemm_bf16_grouped is the form that paid.  The C entry point keeps the mode, tested by a direct call.)
_NUM_CU = 256


def split_k(M: int, N: int, K: int, a_kmaj: bool, b_kmaj: bool) -> int:
    """Parts the contraction of a single-segment product is cut into (1 = no split): outputs of fewer 256 x 256 tiles than
    half the CUs, at least 8 stages (512 deep) per part, tiles x parts <= CUs; with a contraction-contiguous operand the
    number of 64-deep stages has to divide evenly."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    stages = (K + 63) // 64
    if tiles * 2 > _NUM_CU or stages < 16 or os.environ.get("TN_GEMM_VARIANT"):
        return 1
    s = min(_NUM_CU // tiles, stages // 8)
    if not (a_kmaj and b_kmaj):
        while s > 1 and stages % s:
            s -= 1
    return max(s, 1)


def gemm_supported(M: int, N: int, Ks, a_kmaj: bool = False, b_kmaj: bool = False) -> bool:
    """Shapes tn_gemm_bf16 takes (two contraction-major operands: any depth, the tail is zero-filled)."""
    k_ok = all(k > 0 and (k % 64 == 0 or (a_kmaj and b_kmaj)) for k in Ks)
    return k_ok and N % 8 == 0 and M > 0 and (not a_kmaj or M % 8 == 0)


def gemm_tn_supported(M: int, N: int, K: int) -> bool:
    return gemm_supported(M, N, (K,))


def _tn_ok(M: int, K: int, Ns) -> bool:
    return M % 8 == 0 and K % 8 == 0 and all(n % 8 == 0 for n in Ns)


# Which GEMM runs the products of the linear layers below:
#   "own"  the hand-written MFMA kernel of csrc/gemm.hip in its three operand modes — forward x W^T, input gradient
#          dY W (W contraction-major) and weight gradient dY^T x (both contraction-major): every product reads nn.Linear's
#          own tensors, no transposed copy of W, dY or x is made, and the input gradient of a group (q/k/v, gate/up) is ONE
#          multi-segment launch.  Shapes the kernel does not take (K % 64, fewer output tiles than half the CUs) go to the
#          library.
#   "lib"  hipBLASLt through torch, every product brought into the forward layout by transposed copies
#          (tn_transpose_bf16; round 2's default).  Kept for A/B runs: on one box the Qwen2-Audio-7B step takes 742 ms
#          on "own" and 717 ms on "lib" (profiles/r03f_*): the library's kernel runs 12 % faster (same cycles within
#          4 %, higher clock: 2/3 of our LDS read traffic), "own" saves the transposes and their extra stores.
# TN_LINEAR_GEMM / `bench.py --linear-gemm` select; tests/test_kernels_gpu.py holds the two to each other.
LINEAR_GEMM = os.environ.get("TN_LINEAR_GEMM", "own")
_OWN_MIN_TILES = 96          # below this many 256 x 256 output tiles most CUs would idle: the library's split-K wins


def _own(M: int, N: int, Ks, a_kmaj: bool = False, b_kmaj: bool = False) -> bool:
    """The hand-written kernel takes this product (and is the configured choice)."""
    if not (LINEAR_GEMM == "own" and gemm_supported(M, N, Ks, a_kmaj, b_kmaj)):
        return False
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles >= _OWN_MIN_TILES:
        return True
    # few tiles: taken when the contraction can be split over enough units to fill the chip
    return SPLIT_K and len(Ks) == 1 and tiles * split_k(M, N, Ks[0], a_kmaj, b_kmaj) >= _OWN_MIN_TILES


def _bf16_rows(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0
               and t.data_ptr() % 16 == 0 for t in ts)


def _mm_tn(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           accumulate: bool = False, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``a @ b.T (+ bias)`` (``out += ...`` if accumulate; ``+ addend`` [M, N] if given) for contraction-contiguous
    a [M, K], b [N, K]."""
    M, K = a.shape
    N = b.shape[0]
    if addend is not None:
        # (a product that would be cut along its contraction — few output tiles, the last layer on the labelled rows — keeps
        #  its split-K form and gets the addition behind it: the epilogue variant is for whole-tile products)
        if (RESIDUAL_EPILOGUE and _own(M, N, (K,)) and _bf16_rows(a, b, addend) and (bias is None or bias.dtype == torch.bfloat16)
                and K % 64 == 0 and not os.environ.get("TN_GEMM_VARIANT")
                and (not SPLIT_K or split_k(M, N, K, False, False) == 1)):
            return gemm([(a, b)], bias=bias, addend=addend)
        return _mm_tn(a, b, bias) + addend           # (same bits: the product is rounded before the addition either way)
    if (_own(M, N, (K,)) and _bf16_rows(a, b) and (bias is None or bias.dtype == torch.bfloat16)):
        return gemm([(a, b)], bias=bias, out=out, accumulate=accumulate)
    if accumulate:
        return out.addmm_(a, b.t())
    if out is not None:
        return torch.mm(a, b.t(), out=out) if bias is None else torch.addmm(bias, a, b.t(), out=out)
    return torch.nn.functional.linear(a, b, bias)


def _dgrad(dys, ws) -> Optional[torch.Tensor]:
    """sum_i dY_i W_i on the hand-written kernel (W_i [N_i, K] read contraction-major; one launch for up to three
    layers), or None when it does not take the shapes."""
    M, K = dys[0].shape[0], ws[0].shape[1]
    if not (_own(M, K, [w.shape[0] for w in ws]) and _bf16_rows(*dys, *ws)):
        return None
    dx = None
    for i in range(0, len(ws), 3):
        segs = list(zip(dys[i:i + 3], ws[i:i + 3]))
        dx = gemm(segs, b_kmaj=True, out=dx, accumulate=dx is not None)
    return dx


# `_mm_tn(addend=...)` / `gemm(addend=...)`: C = bf16(A B^T + bias) + addend in the GEMM's epilogue (tn_gemm_bf16_addend).  Built
# in round 6 to add the residual stream where o_proj / down_proj / out_proj / fc2 produce their output, bit-identical to
# GEMM + the norm kernel's residual add, and MEASURED A NET LOSS on the headline step (profiles/r06g_*, same box): the norm
# kernels got 6.3 ms cheaper (one input), the epilogue's read of the residual tile is exposed in a persistent GEMM (+53 us per
# launch, +6.8 ms), and the residual gradient — which the fused norm backward adds in passing — became a separate addition
# per block (+6.0 ms).  The models do not use it; the entry point stays in the library, tested.
RESIDUAL_EPILOGUE = True

# bias gradients dY.sum(0) from the weight-gradient launch itself (EPI_BIASG, csrc/gemm.hip) instead of a column-sum pass
BIAS_IN_WGRAD = os.environ.get("TN_BIAS_IN_WGRAD", "1") != "0"          # (A/B switch)


def _wgrad(dy, x2, bias_grad=None) -> Optional[torch.Tensor]:
    """dY^T x [N, K] on the hand-written kernel (both operands contraction-major: the contraction runs over tokens),
    or None.  ``bias_grad`` [N] bf16: filled with dY.sum(0) by the same launch."""
    M, N = dy.shape
    if not (_own(N, x2.shape[1], (M,), True, True) and _bf16_rows(dy, x2)):
        return None
    return gemm([(dy, x2)], True, True, bias_grad=bias_grad)


# Weight-gradient GEMMs on a side stream.  Nothing in the backward consumes a weight gradient, so these products can
# run beside the input-gradient chain: with one workgroup per tile (TN_GEMM_PERSIST=0) the tiles of two launches interleave
# and the last, partly filled round of one kernel (the MLP weight gradients are 688 tiles for 256 CUs: 2.69 rounds) is
# filled by the other's.  `enable_wgrad_stream()` switches it on (bench.py --wgrad-stream / TN_WGRAD_STREAM=1);
# `sync_wgrad_stream()` makes the current stream wait for everything issued there (before the optimizer, before a
# reduce-scatter).  Inputs are `record_stream`ed so the caching allocator does not recycle them under the side stream.
WGRAD_STREAM = None
WGRAD_MAIN = None             # the stream the backward itself runs on (noted by `_beside`)
# A sharding engine (utils/zero_dp.py, FSDP2) reads a returned weight gradient on the backward's own stream right away
# (cast-copy into the reduce-scatter input); it sets this and `_beside` then lets that stream wait for the product.
WGRAD_RETURNS_NEED_SYNC = False


def enable_wgrad_stream(on: bool = True) -> None:
    global WGRAD_STREAM
    # (a stream of another priority than the main one: streams of equal priority may share a hardware queue, and kernels
    #  of one queue never overlap — profiles/r05q_comm_stream_hw_queue_probe.log)
    WGRAD_STREAM = torch.cuda.Stream(priority=int(os.environ.get("TN_WGRAD_STREAM_PRIORITY", "-1"))) if on else None


def wgrad_streams():
    """Streams that may hold unfinished writes of gradients during a backward: the current one, and with the side stream
    on also that one and the backward's own."""
    out = [torch.cuda.current_stream()]
    for st in (WGRAD_STREAM, WGRAD_MAIN if WGRAD_STREAM is not None else None):
        if st is not None and all(st != o for o in out):
            out.append(st)
    return out


def sync_wgrad_stream() -> None:
    if WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(WGRAD_STREAM)


def _beside(inputs, fn, written=()):
    """run `fn` (a weight-gradient product over `inputs`) on the side stream, if there is one.

    `written`: tensors ALLOCATED ON THE MAIN STREAM that `fn` fills on the side stream and that the caller hands to
    autograd next (the bias gradient that rides on a weight-gradient launch): the main stream waits for the side stream
    before it returns, and the allocator is told about the second stream.  A tensor (or list / tuple of tensors) RETURNED
    by `fn` was allocated on the side stream: it is recorded on the main stream, and — when the consumer reads it on the
    main stream right away (`WGRAD_RETURNS_NEED_SYNC`: FSDP2 and every engine without gradient sinks) — waited for."""
    side = WGRAD_STREAM
    if side is None:
        return fn()
    global WGRAD_MAIN
    main = torch.cuda.current_stream()
    WGRAD_MAIN = main
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    for t in inputs:
        t.record_stream(side)
    wait = False
    for t in written:
        if t is not None:
            t.record_stream(side)
            wait = True                           # (read by autograd / the engine's hook on the main stream)
    outs = out if isinstance(out, (list, tuple)) else (out,)
    for t in outs:
        if isinstance(t, torch.Tensor):
            t.record_stream(main)
            wait = wait or WGRAD_RETURNS_NEED_SYNC   # (record_stream only protects the allocator, not the reader)
    if wait:
        main.wait_stream(side)
    return out


# Gradient sinks: a data-parallel engine (utils/zero_dp.py) registers, per weight (by id), an object with
#   take(w) -> (view [N, K] in its reduce-scatter input — fp32 or bf16 —, accumulate?)   and   done(w)
# and the weight-gradient GEMM of that weight writes straight into the view instead of returning a tensor that would have
# to be cast-copied there (autograd then sees no gradient for the weight: the engine counts the arrival itself).
GRAD_SINKS = {}          # id(weight) -> weakref to the engine (a dead engine's entries are ignored)


def _sink_wgrad(w, dy, x2, bias_grad=None) -> bool:
    ref = GRAD_SINKS.get(id(w))
    sink = ref() if ref is not None else None
    if sink is None or not sink.owns(w):
        return False
    M, N = dy.shape
    if not (_own(N, x2.shape[1], (M,), True, True) and _bf16_rows(dy, x2)):
        return False
    view, acc = sink.take(w)
    gemm([(dy, x2)], True, True, out=view, accumulate=acc, bias_grad=bias_grad)
    sink.done(w)
    return True


def _wgrad_group(items):
    """Weight gradients dY_i^T x_i of up to three layers ``items = [(w, dy, x2), ...]`` as ONE grouped launch
    (gemm_grouped_wgrad).  With a data-parallel engine's gradient sinks registered for ALL of them (one dtype, one
    accumulate state) the launch writes the engine's staging views and the result is [None, ...]; with none it returns new
    bf16 tensors; a mixed state falls back to one product per layer."""
    def single(w, dy, x2):
        return None if _sink_wgrad(w, dy, x2) else gemm([(dy, x2)], True, True)
    sinks = []
    for w, _, _ in items:
        ref = GRAD_SINKS.get(id(w))
        sk = ref() if ref is not None else None
        sinks.append(sk if (sk is not None and sk.owns(w)) else None)
    ok = all(_own(dy.shape[1], x2.shape[1], (dy.shape[0],), True, True) and _bf16_rows(dy, x2) for _, dy, x2 in items)
    if not ok or (any(sinks) and not all(sinks)):
        return [single(*it) for it in items]
    pairs = [(dy, x2) for _, dy, x2 in items]
    if not any(sinks):
        return gemm_grouped_wgrad(pairs)
    taken = [sk.take(w) for sk, (w, _, _) in zip(sinks, items)]
    if len({(v.dtype, bool(acc)) for v, acc in taken}) != 1:
        outs = []
        for (view, acc), (w, dy, x2), sk in zip(taken, items, sinks):     # (already taken: write each view by itself)
            gemm([(dy, x2)], True, True, out=view, accumulate=acc)
            sk.done(w)
            outs.append(None)
        return outs
    gemm_grouped_wgrad(pairs, outs=[v for v, _ in taken], accumulate=bool(taken[0][1]))
    for sk, (w, _, _) in zip(sinks, items):
        sk.done(w)
    return [None] * len(items)


def _stacked_view(ts):
    """[n, M, N] view over `ts` if they are equally shaped contiguous [M, N] matrices lying back to back in one storage
    (in order), else None."""
    t0 = ts[0]
    if t0.dim() != 2 or not all(t.shape == t0.shape and t.dtype == t0.dtype and t.is_contiguous() for t in ts):
        return None
    step = t0.numel()
    try:
        base = t0.untyped_storage().data_ptr()
        if not all(t.untyped_storage().data_ptr() == base and t.storage_offset() == t0.storage_offset() + i * step
                   for i, t in enumerate(ts)):
            return None
    except (RuntimeError, NotImplementedError):          # (fake / meta tensors without storage)
        return None
    return t0.as_strided((len(ts), t0.shape[0], t0.shape[1]), (step, t0.shape[1], 1))


class _LinearGroup(torch.autograd.Function):
    """``y_i = x W_i^T + b_i`` for a group of linear layers that share their input (q/k/v, gate/up, or one layer).

    Forward is what nn.Linear does.  LINEAR_GEMM == "own": all three products run on the hand-written kernel in their
    native layouts (see above).  Otherwise the library path of round 2: its WEIGHT gradient dW_i = dY_i^T x contracts
    over tokens, the slow dimension of both operands, which hipBLASLt runs at ~1.0 PFLOP/s on MI355X; there dY_i and x are
    transposed by a HIP kernel (tn_transpose_bf16) and ONE GEMM over the whole group ``[sum N_i, M] x [M, K]`` runs in
    the forward GEMM's layout; the INPUT gradient dX = sum_i dY_i W_i uses W_i transposed first and accumulates over the
    group inside the GEMM epilogue (addmm, beta = 1).
    ``wgrad`` (library path): "tn" (above) | "nt" autograd's layout per layer | "nt_fused" one NT GEMM over the
    column-concatenated dY (the 1280-wide audio tower: three 1280 x 1280 outputs are 25 tiles each).
    ``dgrad_tn=False`` keeps W as stored for the library's input gradient (no gain at 1280 x 1280)."""

    @staticmethod
    def forward(ctx, x, n, wgrad, dgrad_tn, *wb):
        ws, bs = wb[:n], wb[n:]
        src = rope = None
        if isinstance(wgrad, tuple):                       # (wgrad, norm_src, rope): op-level selective recomputation
            wgrad, src, rope = wgrad                       # (`norm_source`) / rotary embedding of outputs (`linear_group`)
        ctx.rope = rope
        if src is not None:
            # x = RMSNorm(h) * w_norm is a ROW kernel's output: keep its input (which the norm's own backward saves
            # anyway) and recompute x in the backward — bit-identical, the kernel norms the stored h
            ctx.save_for_backward(src[0], *ws, src[1])
            ctx.src_eps, ctx.xshape = float(src[2]), x.shape
        else:
            ctx.save_for_backward(x, *ws)
            ctx.src_eps = None
        ctx.has_bias = [b is not None for b in bs]
        ctx.wgrad, ctx.dgrad_tn = wgrad, dgrad_tn
        if LINEAR_GEMM == "own" and x.is_cuda:
            x2 = _c(x.reshape(-1, x.shape[-1]))
            outs = []
            for i, (w, b) in enumerate(zip(ws, bs)):
                if rope is not None and i in rope[3]:
                    # the rotary embedding of this projection in the GEMM's epilogue (or, for shapes the epilogue does not
                    # take, behind it: same bits)
                    cos, sin, D = rope[0], rope[1], rope[2]
                    M_ = x2.shape[0]
                    if (rope_epilogue_ok(M_, w.shape[0], x2.shape[1], D) and _bf16_rows(x2, w)
                            and cos.dtype == torch.bfloat16 and tuple(cos.shape) == (M_, D // 2)):
                        y = gemm_rope(x2, _c(w), b, cos, sin, D)
                    else:
                        y = _mm_tn(x2, _c(w), b)
                        y4 = y.view(1, M_, -1, D)
                        y = L.rope_apply(y4, y4.new_empty(1, M_, 0, D), cos, sin, False)[0].view(M_, -1)
                    outs.append(y.view(*x.shape[:-1], w.shape[0]))
                else:
                    outs.append(_mm_tn(x2, _c(w), b).view(*x.shape[:-1], w.shape[0]))
            return tuple(outs)
        outs = [torch.nn.functional.linear(x, w, b) for w, b in zip(ws, bs)]
        if rope is not None:                               # (library GEMM path / meta tensors: rotate behind the product)
            cos, sin, D = rope[0], rope[1], rope[2]
            for i in rope[3]:
                y4 = outs[i].reshape(1, -1, outs[i].shape[-1] // D, D)
                outs[i] = L.rope_apply(y4, y4.new_empty(1, y4.shape[1], 0, D), cos, sin, False)[0].reshape(outs[i].shape)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x, *ws = ctx.saved_tensors
        if ctx.src_eps is not None:
            norm_w = ws.pop()
            x = L.rmsnorm_fwd(x, None, norm_w, ctx.src_eps)[0].view(ctx.xshape)
        n = len(ws)
        K = x.shape[-1]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        dys = [torch.zeros(M, w.shape[0], dtype=x.dtype, device=x.device) if d is None else _c(d).reshape(M, -1)
               for d, w in zip(dys, ws)]
        if ctx.rope is not None and not ctx.rope[4]:
            # the outputs were rotated: their gradients are rotated back (the transposed rotation) before the products
            # (rope[4]: the consumer did that already — `_AttentionRopeGrad`)
            cos, sin, D, which = ctx.rope[:4]
            which = sorted(which)
            for a in range(0, len(which), 2):
                i, j = which[a], which[a + 1] if a + 1 < len(which) else None
                gi = dys[i].view(1, M, -1, D)
                gj = dys[j].view(1, M, -1, D) if j is not None else gi.new_empty(1, M, 0, D)
                ri, rj = L.rope_apply(gi, gj, cos, sin, True)
                dys[i] = ri.view(M, -1)
                if j is not None:
                    dys[j] = rj.view(M, -1)
        need_x, need_w = ctx.needs_input_grad[0], [ctx.needs_input_grad[4 + i] for i in range(n)]
        Ns = [w.shape[0] for w in ws]
        hip_ok = x.dtype == torch.bfloat16 and (x.is_cuda or x.is_meta) and _tn_ok(M, K, Ns)
        own = LINEAR_GEMM == "own" and x.is_cuda and x.dtype == torch.bfloat16
        dx = None
        if need_x:
            if own:
                dx = _dgrad(dys, [_c(w) for w in ws])
            if dx is not None:
                pass
            elif hip_ok and ctx.dgrad_tn:
                wts = [transpose_2d(_c(w)) for w in ws]                                # W^T [K, N]: forward layout
                dx = _mm_tn(dys[0], wts[0])
                for d, wt in zip(dys[1:], wts[1:]):
                    _mm_tn(d, wt, out=dx, accumulate=True)
                del wts
            else:
                dx = torch.mm(dys[0], ws[0])
                for d, w in zip(dys[1:], ws[1:]):
                    dx.addmm_(d, w)
            dx = dx.view(x.shape)
        dws = [None] * n
        need_b = [hb and ctx.needs_input_grad[4 + n + i] for i, hb in enumerate(ctx.has_bias)]
        dbs = [None] * n
        if any(need_w):
            sunk = set()
            if own:
                x2c = _c(x2)
                for i, (d, nw) in enumerate(zip(dys, need_w)):
                    # the bias gradient rides on the weight-gradient launch (column sums of the dY fragments it reads anyway)
                    bg = (torch.empty(Ns[i], dtype=x.dtype, device=x.device)
                          if (nw and need_b[i] and BIAS_IN_WGRAD and not os.environ.get("TN_GEMM_VARIANT")) else None)
                    if nw and _beside((d, x2c), lambda: _sink_wgrad(ws[i], d, x2c, bg), written=(bg,)):
                        sunk.add(i)
                        dbs[i] = bg
                    elif nw:
                        dws[i] = _beside((d, x2c), lambda: _wgrad(d, x2c, bg), written=(bg,))
                        if dws[i] is not None:
                            dbs[i] = bg
            todo = [i for i in range(n) if need_w[i] and dws[i] is None and i not in sunk]
            if not todo:
                pass
            elif len(todo) < n:                           # (some layers of the group went to the hand-written kernel)
                for i in todo:
                    dws[i] = torch.mm(dys[i].t(), x2)
            elif ctx.wgrad == "nt_fused" and n > 1:
                st = _stacked_view(dys)
                if st is not None:            # the gradients already lie side by side (library.attn_bwd_stacked): no cat
                    dws = list(torch.bmm(st.transpose(1, 2), x2.unsqueeze(0).expand(n, M, K)).unbind(0))
                else:
                    dws = list(torch.split(torch.mm(torch.cat(dys, dim=1).t(), x2), Ns, dim=0))
            elif hip_ok and ctx.wgrad == "tn":
                xt = transpose_2d(_c(x2))                                          # [K, M]
                dyt = torch.empty(sum(Ns), M, dtype=x.dtype, device=x.device)      # [sum N, M]
                o = 0
                for d, N in zip(dys, Ns):
                    transpose_2d(d, out=dyt[o:o + N])
                    o += N
                dw = _mm_tn(dyt, xt)                                               # TN: both contraction-contiguous
                dws = list(torch.split(dw, Ns, dim=0))
            else:
                dws = [torch.mm(d.t(), x2) for d in dys]
            dws = [g if nw else None for g, nw in zip(dws, need_w)]
        dbs = [b if (b is not None or not nb) else column_sum(d) for b, nb, d in zip(dbs, need_b, dys)]
        return (dx, None, None, None, *dws, *dbs)


def norm_source(h, norm_weight, eps):
    """`norm_src` argument of `linear_group` / `swiglu_mlp`: says that their input x IS ``rms_norm(h, norm_weight, eps)`` (h =
    the residual stream the norm saw, i.e. its second output), so that the node may keep h instead of x and recompute x
    in its backward.  This is the op-level selective activation checkpointing of the reference
    (touchnet/models/helper_func.py:39-96: save the outputs of the compute ops — matmuls, attention — and recompute the
    rest) on this path's own autograd nodes: GEMM and attention outputs stay, the ROW kernels' outputs (norm outputs,
    the SwiGLU product) are recomputed."""
    return (h.detach(), norm_weight.detach(), float(eps))


def _norm_src_describes(norm_src, x) -> bool:
    """`norm_src` may stand in for x only if its residual stream has x's rows and dtype; anything else (a sequence-parallel
    wrapper that gathered x to the full sequence behind the norm) silently keeps x itself: a memory optimisation must
    never change or break a step."""
    return (norm_src is not None and tuple(norm_src[0].shape) == tuple(x.shape) and norm_src[0].dtype == x.dtype
            and norm_src[1].shape[-1] == x.shape[-1])


def linear_group(x, layers, wgrad: str = "tn", dgrad_tn: bool = True, norm_src=None, rope=None,
                 rope_grad_in_attention: bool = False):
    """``layers``: list of (weight [N_i, K], bias [N_i] | None) sharing the input ``x`` -> list of outputs.
    ``norm_src``: see `norm_source`.  ``rope = (cos, sin, head_dim, (i, j, ..))``: outputs i, j, .. ([.., heads * head_dim])
    are returned ROTATED — the rotary embedding `apply_rope` would apply to them, computed in the projection's epilogue
    (`gemm_rope`); cos / sin as from `rope_tables`, one row per row of x.  ``rope_grad_in_attention``: the rotated outputs
    go to `packed_attention(.., rope_grad=(cos, sin))` and nowhere else, which returns their gradients rotated back."""
    if wgrad not in ("tn", "nt", "nt_fused"):
        raise ValueError(f"linear_group: wgrad={wgrad!r}")
    if not (x.is_cuda or x.is_meta):
        raise _C.KernelError("linear_group: device tensors only (the product path has no CPU fallback)")
    ws = [w for w, _ in layers]
    bs = [b for _, b in layers]
    if not _norm_src_describes(norm_src, x):
        norm_src = None           # (e.g. under tensor-parallel sequence parallelism x is the gathered sequence): keep x itself
    if rope is not None:
        rope = (rope[0].detach(), rope[1].detach(), int(rope[2]), tuple(int(i) for i in rope[3]),
                bool(rope_grad_in_attention))
    packed = (wgrad, norm_src, rope) if (norm_src is not None or rope is not None) else wgrad
    return list(_LinearGroup.apply(x, len(ws), packed, dgrad_tn, *ws, *bs))


class _SwiGLUMLP(torch.autograd.Function):
    """``down(silu(gate(x)) * up(x))`` as ONE autograd node (modeling_llama.py:174-176, three bias-free nn.Linear).

    LINEAR_GEMM == "own": seven launches of the hand-written kernel (gate, up, down; dW_down, d(act), dX over the
    (gate, up) pair as one two-segment launch, dW_gate, dW_up) + the two SwiGLU kernels; nothing is transposed.
    Library path: same GEMMs in the forward layout, the transposed operands they need — act^T for dW_down,
    [d_gate^T; d_up^T] for the grouped dW_gate/up — written by the SwiGLU kernels themselves (tn_swiglu_fwd_t /
    tn_swiglu_bwd_t: one extra store each) instead of by separate transpose passes."""

    @staticmethod
    def forward(ctx, x, wg, wu, wd, src=None):
        K, I = x.shape[-1], wg.shape[0]
        x2 = _c(x.reshape(-1, K))
        M = x2.shape[0]
        own = _own(M, I, (K,)) and _own(M, K, (I,)) and _own(I, K, (M,), True, True) and _own(K, I, (M,), True, True)
        fused = (own and MLP_EPILOGUE and K % 64 == 0 and _bf16_rows(x2, wg, wu, wd) and wg.stride(0) == wu.stride(0)
                 and not os.environ.get("TN_GEMM_VARIANT"))
        if fused:
            gate, up, act = gemm_swiglu_fwd(x2, wg, wu)       # SwiGLU in the epilogue of ONE gate + up launch
            kept = act
        else:
            gate, up = _mm_tn(x2, _c(wg)), _mm_tn(x2, _c(wu))
        if fused:
            pass
        elif own:
            act = L.swiglu_fwd(gate, up)
            kept = act
        else:
            act, kept = L.swiglu_fwd_t(gate, up)                       # kept = act^T
        y = _mm_tn(act, _c(wd))
        if src is not None and own:
            # op-level selective recomputation (`norm_source`): neither the norm output x nor act = silu(gate) * up is
            # kept — both are row kernels' outputs, recomputed bit-identically in the backward from h and (gate, up)
            ctx.save_for_backward(src[0].reshape(-1, K), gate, up, src[1], wg, wu, wd)
            ctx.src_eps = float(src[2])
        else:
            ctx.save_for_backward(x2, gate, up, kept, wg, wu, wd)
            ctx.src_eps = None
        ctx.xshape, ctx.own, ctx.fused = x.shape, own, fused
        return y.view(*x.shape[:-1], wd.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, gate, up, kept, wg, wu, wd = ctx.saved_tensors
        if ctx.src_eps is not None:
            x2 = L.rmsnorm_fwd(x2, None, kept, ctx.src_eps)[0]           # (x2 held h, `kept` the norm weight)
            kept = L.swiglu_fwd(gate, up)
        M, K = x2.shape
        I, H = wg.shape[0], wd.shape[0]
        dy2 = _c(dy).reshape(M, H)
        nx, ng, nu, nd = ctx.needs_input_grad[:4]
        if ctx.own and LINEAR_GEMM == "own":
            grouped = GROUPED_WGRAD and ng and nu and nd and not os.environ.get("TN_GEMM_VARIANT")
            if not grouped:
                dwd = None if (not nd or _beside((dy2, kept), lambda: _sink_wgrad(wd, dy2, kept))) \
                    else _beside((dy2, kept), lambda: gemm([(dy2, kept)], True, True))                   # dY^T act  [H, I]
            if ctx.fused and H % 64 == 0 and _bf16_rows(dy2, wd):
                dgate, dup = gemm_swiglu_bwd(dy2, wd, gate, up)     # d(act) = dY W_down lives in the accumulators only
            else:
                dact = gemm([(dy2, _c(wd))], b_kmaj=True)                              # dY W_down [M, I]
                dgate, dup = L.swiglu_bwd(dact, gate, up)
                del dact
            dx = gemm([(dgate, _c(wg)), (dup, _c(wu))], b_kmaj=True).view(ctx.xshape) if nx else None
            if grouped:
                # ONE launch for the three weight gradients (3 x 688 tiles = 8 whole rounds + 16 tiles split-K)
                dwg, dwu, dwd = _beside((dgate, dup, dy2, x2, kept), lambda: _wgrad_group(
                    [(wg, dgate, x2), (wu, dup, x2), (wd, dy2, kept)]))
                return dx, dwg, dwu, dwd, None
            dwg = None if (not ng or _beside((dgate, x2), lambda: _sink_wgrad(wg, dgate, x2))) \
                else _beside((dgate, x2), lambda: gemm([(dgate, x2)], True, True))
            dwu = None if (not nu or _beside((dup, x2), lambda: _sink_wgrad(wu, dup, x2))) \
                else _beside((dup, x2), lambda: gemm([(dup, x2)], True, True))
            return dx, dwg, dwu, dwd, None
        act_t = kept if not ctx.own else transpose_2d(kept)
        dwd = _mm_tn(transpose_2d(dy2), act_t) if nd else None                      # [H, I], forward layout
        dact = _mm_tn(dy2, transpose_2d(_c(wd)))                                    # [M, I]
        dgate, dup, dgu_t = L.swiglu_bwd_t(dact, gate, up)
        del dact
        dx = None
        if nx:
            dx = _mm_tn(dgate, transpose_2d(_c(wg)))
            _mm_tn(dup, transpose_2d(_c(wu)), out=dx, accumulate=True)
            dx = dx.view(ctx.xshape)
        dwg = dwu = None
        if ng or nu:
            dwg, dwu = torch.split(_mm_tn(dgu_t, transpose_2d(x2)), [I, I], dim=0)
        return dx, dwg if ng else None, dwu if nu else None, dwd, None


class _Conv1dK3(torch.autograd.Function):
    """``conv1d(kernel 3, padding 1, stride s)`` over channels-last sequences as GEMMs of the hand-written kernel — the
    Whisper / Qwen2-Audio conv stem (transformers' WhisperEncoder.conv1 / conv2 behind touchnet/models/qwen2_audio/
    __init__.py:52-73), which the reference runs through cuDNN/MIOpen.

    x [n, T, C] is copied once into zero-separated slots of P = T + 2 rows ([0, x_0 .. x_{T-1}, 0]; P a multiple of s) of
    ONE long [n P + s, C] sequence.  The im2col matrix then needs no copy: row r = the 3 C contiguous elements from row
    r s on — a strided VIEW with overlapping rows (pitch s C < 3 C), which the GEMM's DMA descriptors read as they are:
        forward      Y [n P / s, O] = A W2^T + b          W2[o, k C + c] = w[o, c, k]
        dW2          = dY^T A        (both contraction-major: the contraction runs over the output rows)
        dX           = overlap-add of dA = dY W2 over the three taps (fp32 accumulation, one rounding)
    Of the P / s output rows of a slot the first T_out are the convolution, the last one(s) mix two clips: the result is
    returned WHOLE ([n, P / s, O]; the caller slices behind its activation) so that nothing is copied here."""

    @staticmethod
    def forward(ctx, x, w, b, stride, need_dx):
        n, T, C0 = x.shape
        O, s = w.shape[0], int(stride)
        C = (C0 + 63) // 64 * 64                 # channels padded with zeros: 3 C is then a multiple of the 64-deep stage
        P = (T + 2 + s - 1) // s * s
        R = n * P // s
        xp = x.new_zeros(n * P + s + 2, C)                                 # (+ tail rows: the last window stays inside)
        xp[:n * P].view(n, P, C)[:, 1:T + 1, :C0] = x
        A = xp.as_strided((R, 3 * C), (s * C, 1))
        w2 = w.new_zeros(O, 3, C)
        w2[:, :, :C0] = w.permute(0, 2, 1)
        w2 = w2.view(O, 3 * C)
        y = gemm([(A, w2)], bias=b)
        ctx.save_for_backward(xp, w2)
        ctx.geom = (n, T, C0, C, O, s, P, R, bool(need_dx), b is not None, w.dtype)
        return y.view(n, P // s, O)

    @staticmethod
    def backward(ctx, dy):
        xp, w2 = ctx.saved_tensors
        n, T, C0, C, O, s, P, R, need_dx, has_b, wdtype = ctx.geom
        dyf = _c(dy).reshape(R, O)
        A = xp.as_strided((R, 3 * C), (s * C, 1))
        dw2 = gemm([(dyf, A)], True, True)                                 # [O, 3 C]
        dw = dw2.view(O, 3, C)[:, :, :C0].permute(0, 2, 1).contiguous().to(wdtype)
        db = column_sum(dyf) if has_b else None
        dx = None
        if need_dx:
            dA = gemm([(dyf, w2)], b_kmaj=True)                            # [R, 3 C]
            acc = torch.zeros(n * P + s + 2, C, dtype=torch.float32, device=dy.device)
            for k in range(3):                                             # tap k of output row r lands on row r s + k
                acc[k:k + R * s:s] += dA[:, k * C:(k + 1) * C]
            dx = acc[:n * P].view(n, P, C)[:, 1:T + 1, :C0].to(dy.dtype)
        return dx, dw, db, None, None


def conv1d_k3(x, weight, bias, stride: int = 1, need_dx: bool = True):
    """x [n, T, C] (channels last) -> (y [n, P / stride, O] with the convolution in rows [0, T_out), T_out); see
    _Conv1dK3.  bf16 device tensors, output channels a multiple of 8."""
    n, T, C = x.shape
    if not (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and weight.shape[2] == 3 and weight.shape[1] == C and weight.shape[0] % (64 if need_dx else 8) == 0):
        raise _C.KernelError("conv1d_k3: bf16 device tensors, kernel size 3, output channels a multiple of 8 (of 64 when the "
                             "input gradient is wanted: it contracts over them)")
    return _Conv1dK3.apply(_c(x), weight, bias, stride, need_dx), (T + 2 - 3) // stride + 1


_MLP_FUSED = os.environ.get("TN_MLP_FUSED", "1") != "0"       # (A/B switch for measurements)


GELU_EPILOGUE = os.environ.get("TN_GELU_EPILOGUE", "1") != "0"          # (A/B switch; same bits either way)


class _GeluMLP(torch.autograd.Function):
    """``fc2(gelu(fc1(x)))`` of the Whisper-style encoder layer as ONE autograd node (round 6): GELU in fc1's epilogue (pre
    and act from one launch), its backward in the epilogue of fc2's input-gradient product (d(act) stays in the
    accumulators); weight and bias gradients as `_LinearGroup` forms them (bias gradients ride on the weight-gradient
    launches).  Every launch is bit-identical to the kernels it replaces.  Same box, interleaved: 707.5 -> 705.5 ms per
    headline step (profiles/r06g_*: the two GELU passes, 9.3 ms, against 4.9 ms of epilogue time inside the GEMMs)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        K, I, H = x.shape[-1], w1.shape[0], w2.shape[0]
        x2 = _c(x.reshape(-1, K))
        pre, act = gemm_gelu_fwd(x2, _c(w1), b1)
        y = _mm_tn(act, _c(w2), b2)
        ctx.save_for_backward(x2, pre, act, w1, w2)
        ctx.xshape, ctx.has_b = x.shape, (b1 is not None, b2 is not None)
        return y.view(*x.shape[:-1], H)

    @staticmethod
    def backward(ctx, dy):
        x2, pre, act, w1, w2 = ctx.saved_tensors
        M, K = x2.shape
        I, H = w1.shape[0], w2.shape[0]
        dy2 = _c(dy).reshape(M, H)
        nx, n1, nb1, n2, nb2 = ctx.needs_input_grad[:5]
        dpre = gemm_gelu_bwd(dy2, _c(w2), pre)                                   # (dY W2) o gelu'(pre)   [M, I]
        dx = gemm([(dpre, _c(w1))], b_kmaj=True).view(ctx.xshape) if nx else None

        def wgrad(w, d, a, want_w, want_b):
            bg = (torch.empty(d.shape[1], dtype=d.dtype, device=d.device)
                  if (want_w and want_b and BIAS_IN_WGRAD) else None)
            dw = None
            if want_w:
                if _beside((d, a), lambda: _sink_wgrad(w, d, a, bg), written=(bg,)):
                    dw = None
                else:
                    dw = _beside((d, a), lambda: _wgrad(d, a, bg), written=(bg,))
                    if dw is None:                                             # (a shape the kernel does not take)
                        dw, bg = torch.mm(d.t(), a), None
            db = bg if bg is not None else (column_sum(d) if want_b else None)
            return dw, db
        dw2, db2 = wgrad(w2, dy2, act, n2, nb2 and ctx.has_b[1])
        dw1, db1 = wgrad(w1, dpre, x2, n1, nb1 and ctx.has_b[0])
        return dx, dw1, db1, dw2, db2


def gelu_mlp(x, w1, b1, w2, b2):
    """The Whisper / Qwen2-Audio encoder layer's MLP ``fc2(gelu(fc1(x)))``: bf16 device tensors of shapes the GELU
    epilogues take go through the fused node above, anything else composes the individual ops (same bits)."""
    if not (x.is_cuda or x.is_meta):
        raise _C.KernelError("gelu_mlp: device tensors only (the product path has no CPU fallback)")
    M, K = x.numel() // x.shape[-1], x.shape[-1]
    I, H = w1.shape[0], w2.shape[0]
    ok = (GELU_EPILOGUE and x.is_cuda and LINEAR_GEMM == "own" and x.dtype == torch.bfloat16 and K % 64 == 0 and I % 64 == 0
          and _own(M, I, (K,)) and _own(M, H, (I,)) and _own(M, I, (H,), False, True) and _own(M, K, (I,), False, True)
          and _bf16_rows(w1, w2) and all(b is None or b.dtype == torch.bfloat16 for b in (b1, b2))
          and not os.environ.get("TN_GEMM_VARIANT") and os.environ.get("TN_GEMM_M16", "1") != "0")
    if ok:
        return _GeluMLP.apply(x, w1, b1, w2, b2)
    h = linear_group(x, [(w1, b1)], wgrad="nt", dgrad_tn=False)[0]
    return linear_group(gelu(h), [(w2, b2)], wgrad="nt", dgrad_tn=False)[0]


def swiglu_mlp(x, w_gate, w_up, w_down, norm_src=None):
    """Llama/Qwen2 MLP ``down(silu(gate(x)) * up(x))`` (bias-free).  bf16 device tensors with 8-aligned shapes take the
    fused node above; anything else composes the individual ops (same maths).  ``norm_src``: see `norm_source`."""
    if not (x.is_cuda or x.is_meta):
        raise _C.KernelError("swiglu_mlp: device tensors only (the product path has no CPU fallback)")
    M = x.numel() // x.shape[-1]
    if not _norm_src_describes(norm_src, x):
        norm_src = None
    if (_MLP_FUSED and x.dtype == torch.bfloat16
            and all(w.dtype == torch.bfloat16 for w in (w_gate, w_up, w_down))
            and _tn_ok(M, x.shape[-1], (w_gate.shape[0], w_down.shape[0]))):
        return _SwiGLUMLP.apply(x, w_gate, w_up, w_down, norm_src)
    gate, up = linear_group(x, [(w_gate, None), (w_up, None)], norm_src=norm_src)
    return linear_group(swiglu(gate, up), [(w_down, None)])[0]

# ------------------------------------------------------------------------------------ frontend
_MEL_CACHE = {}


def slaney_mel_filters(n_mels: int, device) -> torch.Tensor:
    """librosa.filters.mel(sr=16000, n_fft=400, n_mels) (touchnet/data/functions.py:179-182): slaney scale
    and norm.  A constant table, built on the host once per (n_mels, device)."""
    key = (n_mels, str(device))
    if key not in _MEL_CACHE:
        import numpy as np
        sr, n_fft = 16000.0, 400
        f_sp, min_log_hz = 200.0 / 3, 1000.0
        min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
        to_mel = lambda f: np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep,
                                    f / f_sp)
        to_hz = lambda m: np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
        fftfreqs = np.linspace(0.0, sr / 2, 1 + n_fft // 2)
        mel_f = to_hz(np.linspace(to_mel(np.float64(0.0)), to_mel(np.float64(sr / 2)), n_mels + 2))
        fdiff = np.diff(mel_f)
        ramps = mel_f[:, None] - fftfreqs[None, :]
        lower = -ramps[:-2] / fdiff[:-1, None]
        upper = ramps[2:] / fdiff[1:, None]
        w = np.maximum(0.0, np.minimum(lower, upper)) * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]
        _MEL_CACHE[key] = torch.from_numpy(w.astype(np.float32)).to(device).contiguous()
    return _MEL_CACHE[key]


def kaldi_fbank(wav: torch.Tensor, num_mel_bins: int = 80) -> torch.Tensor:
    """wav fp32 [N] in [-1, 1) at 16 kHz -> fp32 [frames, num_mel_bins] (functions.py:117-134)."""
    return L.kaldi_fbank(_c(wav).float().view(-1), int(num_mel_bins))


def log_mel_spectrogram(wav: torch.Tensor, num_mel_bins: int = 128, padding: int = 0) -> torch.Tensor:
    """wav fp32 [N] at 16 kHz -> fp32 [N // 160, num_mel_bins] (functions.py:159-190)."""
    wav = _c(wav).float().view(-1)
    if padding > 0:
        wav = torch.nn.functional.pad(wav, (0, padding))
    return L.log_mel(wav, slaney_mel_filters(num_mel_bins, wav.device), int(num_mel_bins))


def pcm16_to_float(pcm: torch.Tensor) -> torch.Tensor:
    """int16 PCM on the device -> float32 in [-1, 1) (x / 32768, exact; touchnet/data/datapipe.py:164)."""
    if not pcm.is_cuda or pcm.dtype != torch.int16:
        raise RuntimeError("pcm16_to_float: expects an int16 device tensor")
    return L.pcm16_to_f32(_c(pcm))


def bestrq_tokenize(feat: torch.Tensor, quantizer: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
    """BEST-RQ codes (touchnet/tokenizer/tokenizer.py:289-299): feat [T, F], quantizer [F, E], L2-normalised
    codebook [V, E], all fp32 on the device -> int64 [T]."""
    for t in (feat, quantizer, codebook):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
            raise RuntimeError("bestrq_tokenize: expects 2-D fp32 device tensors")
    T, Fdim = feat.shape
    E, V = quantizer.shape[1], codebook.shape[0]
    if quantizer.shape[0] != Fdim or codebook.shape[1] != E:
        raise RuntimeError(f"bestrq_tokenize: shape mismatch feat {tuple(feat.shape)} quantizer {tuple(quantizer.shape)} "
                           f"codebook {tuple(codebook.shape)}")
    if E not in (8, 16, 32):
        raise _C.KernelError(f"bestrq_tokenize: codebook width {E} not in (8, 16, 32)")
    return L.bestrq_tokenize(_c(feat), _c(quantizer), _c(codebook))


SPEED_ZEROS = 32            # zero crossings of the interpolation kernel on each side (at the cutoff's scale)
SPEED_ROLLOFF = 0.95        # pass band as a fraction of the narrower Nyquist band
SPEED_BETA = 14.769656459379492   # Kaiser window: ~ -150 dB side lobes
_SPEED_TABLES = {}


def speed_filter_bank(speed: float):
    """(p, q, table [q, ntap] float64) of the polyphase band-limited interpolator for `speed` = p / q:
    h(t) = c sinc(c t) kaiser(t / W), c = rolloff * min(1, 1 / speed), W = zeros / c input samples; row r holds
    h(r / q - (j - ntap / 2 + 1)) for j = 0 .. ntap - 1."""
    import fractions

    import numpy as np
    fr = fractions.Fraction(str(speed)).limit_denominator(1000)
    p_, q_ = fr.numerator, fr.denominator
    c = SPEED_ROLLOFF * min(1.0, q_ / p_)
    W = SPEED_ZEROS / c
    ntap = 2 * int(math.ceil(W))
    j = np.arange(ntap, dtype=np.float64)[None, :] - (ntap // 2 - 1)          # input offset k - i
    t = np.arange(q_, dtype=np.float64)[:, None] / q_ - j                     # (n s - k)
    win = np.where(np.abs(t) < W, np.i0(SPEED_BETA * np.sqrt(np.clip(1.0 - (t / W) ** 2, 0.0, None))) / np.i0(SPEED_BETA), 0.0)
    return p_, q_, c * np.sinc(c * t) * win


def speed_perturb(wave: torch.Tensor, speed: float) -> torch.Tensor:
    """fp32 waveform [N] on the device played `speed` times faster (pitch and tempo), same sample rate:
    floor(N / speed) samples (touchnet/data/functions.py:99-114: sox `speed` + `rate`)."""
    if not wave.is_cuda or wave.dim() != 1:
        raise RuntimeError("speed_perturb: expects a 1-D device waveform")
    if speed == 1.0:
        return wave
    key = (float(speed), wave.device)
    if key not in _SPEED_TABLES:
        p_, q_, tab = speed_filter_bank(speed)
        _SPEED_TABLES[key] = (p_, q_, torch.from_numpy(tab).to(torch.float32).to(wave.device).contiguous())
    p_, q_, tab = _SPEED_TABLES[key]
    n_out = (wave.numel() * q_) // p_
    return L.resample_polyphase(_c(wave).float(), tab, p_, q_, max(n_out, 1))


def feat_augment(feat: torch.Tensor, t_masks=(), f_masks=(), subs=(), out_rows: Optional[int] = None) -> torch.Tensor:
    """fp32 [T, F] -> fp32 [out_rows, F]: zero stripes (spec_aug), row substitutions from earlier rows (spec_sub) and the
    tail trim (spec_trim) of touchnet/data/functions.py:193-255 in one pass; the draws come from the caller."""
    if not feat.is_cuda or feat.dim() != 2:
        raise RuntimeError("feat_augment: expects a 2-D device tensor")
    T = feat.shape[0]
    return L.feat_augment(_c(feat).float(), list(t_masks), list(f_masks), list(subs), T if out_rows is None else int(out_rows))


def audiofeat_stack(feat: torch.Tensor, stack: int, stride: int, normalize: bool = True) -> torch.Tensor:
    """feat fp32 [T, F] -> fp32 [ceil(T/stride), F*stack] (functions.py:258-286)."""
    return L.audiofeat_stack(_c(feat).float(), int(stack), int(stride), bool(normalize))
