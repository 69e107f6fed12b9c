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


def fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence, ignore_index: int = -100,
                               chunk_tokens: int = 16384, compact: bool = False):
    """(loss_per_sample, loss_per_token, accuracy) straight from the final hidden states: lm_head GEMM +
    packed CE chunked over tokens (touchnet_amd.functional.fused_linear_cross_entropy)."""
    loss, stats = ops().fused_linear_cross_entropy(hidden, weight, labels, sentence_lens, num_sentence,
                                                   ignore_index, chunk_tokens, compact)
    return loss, stats[1], stats[2]
