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

import weakref

import torch

from touchnet_amd.models.backend import ops

# accuracy of the last loss_fn call, valid for exactly the tensor OBJECTS it was computed from (weak references +
# in-place version counters: an address recycled by the caching allocator or a tensor modified in place can never
# hit) and for one acc_fn call (cleared on the hit)
_LAST = {"pred": None, "labels": None, "versions": None, "acc": None}


def _remember(pred, labels, acc):
    _LAST.update(pred=weakref.ref(pred), labels=weakref.ref(labels), versions=(pred._version, labels._version), acc=acc)


def cross_entropy_loss(pred, labels, sentence_lens, num_sentence, ignore_index: int = -100):
    loss, stats = ops().packed_cross_entropy(pred, labels, sentence_lens, num_sentence, ignore_index)
    _remember(pred, labels, stats[2])
    return loss, stats[1]


def cached_accuracy(pred, labels):
    hit = (_LAST["pred"] is not None and _LAST["pred"]() is pred and _LAST["labels"]() is labels
           and _LAST["versions"] == (pred._version, labels._version))
    acc = _LAST["acc"] if hit else None
    _LAST.update(pred=None, labels=None, versions=None, acc=None)
    return acc


def fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence, ignore_index: int = -100,
                               chunk_tokens: int = 4096, compact: bool = False, tp=None):
    """(loss_per_sample, loss_per_token, accuracy) straight from the final hidden states: lm_head GEMM +
    packed CE chunked over tokens (touchnet_amd.functional.fused_linear_cross_entropy).  `tp = (group, rank, size)`: the
    head is vocabulary-sharded over the tensor-parallel group (loss parallel, touchnet/loss/cross_entropy.py:29-33)."""
    if tp is None:
        loss, stats = ops().fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence,
                                                       ignore_index, chunk_tokens, compact)
    else:
        loss, stats = ops().fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence,
                                                       ignore_index, chunk_tokens, compact, tp=tp)
    return loss, stats[1], stats[2]
