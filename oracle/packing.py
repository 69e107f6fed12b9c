"""Oracle: greedy first-fit sequence packers (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates
  * ``batch_text``                       touchnet/models/llama/processing_llama.py:24-104
  * ``batch_pairaudio_pairtext_packed``  touchnet/models/touch_audio/processing_touch_audio.py:117-214
as plain Python loops over numpy int64 buffers.  Bit-exact integer outputs are
required (checked against tests/golden/packing_*.npz, generated from the reference).
"""
from __future__ import annotations

import numpy as np


def _fresh(B, T, pad_id, feat_dim=None):
    buf = {
        "input_ids": np.full((B, T), pad_id, dtype=np.int64),
        "labels": np.full((B, T), -100, dtype=np.int64),
        "position_ids": np.zeros((B, T), dtype=np.int64),
        "attention_mask": np.zeros((B, T), dtype=np.int64),   # document ids, 0 = pad
        "sentence_lens": np.ones((B, T), dtype=np.int64),     # 1 on pad (avoid /0)
        "num_sentence": 0,
    }
    if feat_dim is not None:
        buf["input_features"] = np.zeros((B, T, feat_dim), dtype=np.float32)
    return buf


def batch_text(samples, batchsize, seqlen, bos, eos, pad, drop_last=False):
    """processing_llama.py:24-104.  ``samples`` = iterable of token-id lists.

    Row switch / yield rules (processing_llama.py:63-89): a sentence of
    ``len+1`` slots that does not fit the current row moves to the next row; if the
    current row is the last one the buffer is emitted first.  The reference never
    checks that a sentence fits an empty row (over-long text is filtered upstream,
    touchnet/data/functions.py:52-80), and neither do we.
    """
    buf = _fresh(batchsize, seqlen, pad)
    row, col, sent = 0, 0, 1
    for ids in samples:
        n = len(ids) + 1
        if col + n > seqlen:
            if row == batchsize - 1:
                yield buf
                buf = _fresh(batchsize, seqlen, pad)
                row, col, sent = 0, 0, 1
            else:
                row, col, sent = row + 1, 0, 1
        sl = slice(col, col + n)
        buf["input_ids"][row, sl] = [bos] + list(ids)
        buf["labels"][row, sl] = list(ids) + [eos]
        buf["position_ids"][row, sl] = np.arange(n)
        buf["attention_mask"][row, sl] = sent
        buf["sentence_lens"][row, sl] = n
        buf["num_sentence"] += 1
        col += n
        sent += 1
    if (not drop_last) and (col > 0 or row > 0):
        yield buf


def batch_pairaudio_pairtext_packed(samples, batchsize, seqlen, feat_dim, bos, eos, pad,
                                    drop_last=False):
    """processing_touch_audio.py:117-214.  ``samples`` = iterable of
    ``(audiofeat float32 [Ta, F], token-id list)``.

    A segment is ``Ta`` audio slots followed by ``len+1`` text slots; samples whose
    segment exceeds ``seqlen`` are dropped (``:169-170``); ``sentence_lens`` is the
    TEXT length over the whole segment (``:207``); labels stay -100 on audio slots.
    """
    buf = _fresh(batchsize, seqlen, pad, feat_dim)
    row, col, sent = 0, 0, 1
    for feat, ids in samples:
        feat = np.asarray(feat, dtype=np.float32)
        ta, nt = feat.shape[0], len(ids) + 1
        tot = ta + nt
        if tot > seqlen:
            continue
        if col + tot > seqlen:
            if row == batchsize - 1:
                buf["shift_labels"] = buf["labels"]
                yield buf
                buf = _fresh(batchsize, seqlen, pad, feat_dim)
                row, col, sent = 0, 0, 1
            else:
                row, col, sent = row + 1, 0, 1
        buf["input_features"][row, col:col + ta] = feat
        tsl = slice(col + ta, col + tot)
        buf["input_ids"][row, tsl] = [bos] + list(ids)
        buf["labels"][row, tsl] = list(ids) + [eos]
        seg = slice(col, col + tot)
        buf["position_ids"][row, seg] = np.arange(tot)
        buf["attention_mask"][row, seg] = sent
        buf["sentence_lens"][row, seg] = nt
        buf["num_sentence"] += 1
        col += tot
        sent += 1
    if (not drop_last) and (row > 0 or col > 0):
        buf["shift_labels"] = buf["labels"]
        yield buf
