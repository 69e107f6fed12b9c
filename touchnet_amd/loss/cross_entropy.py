"""TrainSpec `loss_fn` for packed batches — same signature and return contract as
touchnet/loss/cross_entropy.py:12-50:

    loss_fn(pred[B,T,V], labels[B,T], sentence_lens[B,T], num_sentence, ignore_index=-100)
        -> (loss_per_sample  [0-d, differentiable: backward target],
            loss_per_token   [0-d, logging])

executed by one HIP kernel pass over the logits (no fp32 upcast copy, no `.item()` host syncs:
the reference does two, cross_entropy.py:35,41).  The argmax accuracy of the same pass is cached so the
TrainSpec `acc_fn` (touchnet/utils/metrics.py:26-50) does not re-read the logits.
"""
from __future__ import annotations

import torch

from touchnet_amd.models.backend import ops

_LAST = {"key": None, "acc": None}


def _key(pred, labels):
    return (pred.data_ptr(), tuple(pred.shape), labels.data_ptr())


def cross_entropy_loss(pred, labels, sentence_lens, num_sentence, ignore_index: int = -100):
    loss, stats = ops().packed_cross_entropy(pred, labels, sentence_lens, num_sentence, ignore_index)
    _LAST["key"], _LAST["acc"] = _key(pred, labels), stats[2]
    return loss, stats[1]


def cached_accuracy(pred, labels):
    if _LAST["key"] == _key(pred, labels):
        return _LAST["acc"]
    return None


class _FusedLinearCE(torch.autograd.Function):
    """lm_head GEMM + packed CE, chunked over tokens so the [B*T, V] logits never exist at once
    (the liger fused-linear-CE idea, touchnet/bin/train.py:443-445 — but keeping the reference's
    per-sentence normalisation, which liger's mean-over-tokens drops, SURVEY.md §2.3 K10').
    Gradients w.r.t. hidden and weight are produced in the forward pass and scaled by the upstream
    gradient in backward."""

    @staticmethod
    def forward(ctx, hidden, weight, labels, sentence_lens, num_sentence, ignore_index, chunk):
        from touchnet_amd import _C
        F = ops()
        H = hidden.shape[-1]
        h2 = hidden.reshape(-1, H)
        n, V = h2.shape[0], weight.shape[0]
        lab = labels.reshape(-1).to(torch.int64).contiguous()
        sl = sentence_lens.reshape(-1).to(torch.int64).contiguous()
        ns = F._num_sentence_dev(num_sentence, hidden.device)
        one = torch.ones(1, dtype=torch.float32, device=hidden.device)
        nll = torch.empty(n, dtype=torch.float32, device=hidden.device)
        lse = torch.empty_like(nll)
        hit = torch.empty(n, dtype=torch.int32, device=hidden.device)
        out = torch.empty(4, dtype=torch.float32, device=hidden.device)
        dh = torch.empty_like(h2)
        dw = None
        lib, p, st = _C.lib(), _C.ptr, _C.stream
        for s in range(0, n, chunk):
            e = min(s + chunk, n)
            logits = torch.nn.functional.linear(h2[s:e], weight)              # [c, V]
            _C.check(lib.tn_ce_forward(p(logits), p(lab[s:e]), p(sl[s:e]), p(ns), p(nll[s:e]), p(lse[s:e]),
                                       p(hit[s:e]), None, e - s, V, int(ignore_index), _C.dcode(logits), st()),
                     "tn_ce_forward")
            _C.check(lib.tn_ce_backward(p(logits), p(logits), p(lab[s:e]), p(sl[s:e]), p(lse[s:e]), p(ns), p(one),
                                        e - s, V, int(ignore_index), _C.dcode(logits), st()), "tn_ce_backward")
            torch.mm(logits, weight, out=dh[s:e])                             # dh = dlogits @ W
            if dw is None:
                dw = torch.mm(logits.t(), h2[s:e])                            # dW = dlogits^T @ h
            else:
                dw.addmm_(logits.t(), h2[s:e])                                # fp32 accumulate inside the GEMM
            del logits
        _C.check(lib.tn_ce_reduce(p(nll), p(hit), p(lab), p(sl), p(ns), p(out), n, int(ignore_index), st()),
                 "tn_ce_reduce")
        ctx.save_for_backward(dh, dw)
        ctx.hshape, ctx.wdtype = hidden.shape, weight.dtype
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_loss, _g):
        dh, dw = ctx.saved_tensors
        g = g_loss.to(torch.float32)
        return (dh * g.to(dh.dtype)).view(ctx.hshape), (dw * g).to(ctx.wdtype), None, None, None, None, None


def fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence, ignore_index: int = -100,
                               chunk_tokens: int = 16384):
    """(loss_per_sample, loss_per_token, accuracy) straight from the final hidden states."""
    loss, stats = _FusedLinearCE.apply(hidden, weight, labels, sentence_lens, num_sentence, ignore_index,
                                       chunk_tokens)
    return loss, stats[1], stats[2]
