"""Sequence packers of the TouchAudio recipes (datapipe stages, same signatures as the reference's).

`batch_audio_packed(data, config, tokenizer)` (audio pretraining, BEST-RQ labels), bit-identical to
touchnet/models/touch_audio/processing_touch_audio.py:25-114, and
`batch_pairaudio_pairtext_packed(data, config, tokenizer)` (ASR pairs), bit-identical to :117-214:

Segment = `audio_len` feature frames followed by `len(ids)+1` text slots.  input_features carries the
frames (zeros elsewhere), input_ids is pad on audio slots and bos+ids on text slots, labels are -100 on
audio slots and ids+eos on text slots, position_ids run over the whole segment, sentence_lens holds the
TEXT length over the whole segment, `shift_labels` aliases `labels`, samples longer than a row are dropped.
"""
from __future__ import annotations

import numpy as np
import torch

from ..llama.processing_llama import PackBuffer


def _emit_asr(buf: PackBuffer, feats, sentences, feat_dim, bos, eos, pad):
    B, T = buf.B, buf.T
    input_ids = np.full(B * T, pad, dtype=np.int64)
    labels = np.full(B * T, -100, dtype=np.int64)
    position_ids = np.zeros(B * T, dtype=np.int64)
    attention_mask = np.zeros(B * T, dtype=np.int64)
    sentence_lens = np.ones(B * T, dtype=np.int64)
    dev = feats[0].device if feats else torch.device("cpu")   # a device-side frontend keeps features in HBM
    input_features = torch.zeros(B, T, feat_dim, dtype=torch.float32, device=dev)
    if len(buf):
        seg, within, flat = buf.scatter_index()
        lens = np.asarray(buf.lens, dtype=np.int64)
        tlen = np.asarray([len(s) + 1 for s in sentences], dtype=np.int64)
        alen = lens - tlen
        position_ids[flat] = within
        attention_mask[flat] = np.asarray(buf.sents, dtype=np.int64)[seg]
        sentence_lens[flat] = tlen[seg]
        tpos = within - alen[seg]                          # index inside the text part (<0 on audio slots)
        is_text = tpos >= 0
        toks = np.concatenate([np.asarray(s, dtype=np.int64) for s in sentences])
        inp = np.full(flat.size, pad, dtype=np.int64)
        lab = np.full(flat.size, -100, dtype=np.int64)
        head = is_text & (tpos == 0)
        tail = is_text & (tpos == tlen[seg] - 1)
        inp[head] = bos
        inp[is_text & ~head] = toks
        lab[tail] = eos
        lab[is_text & ~tail] = toks
        input_ids[flat], labels[flat] = inp, lab
        for r, c, f in zip(buf.rows, buf.cols, feats):
            input_features[r, c:c + f.shape[0]] = f
    t = lambda a: torch.from_numpy(a.reshape(B, T))
    lab_t = t(labels)
    return {"input_ids": t(input_ids), "input_features": input_features, "labels": lab_t,
            "position_ids": t(position_ids), "attention_mask": t(attention_mask),
            "sentence_lens": t(sentence_lens), "num_sentence": len(buf), "shift_labels": lab_t,
            "labelled_rows_max": int(sum(len(s) + 1 for s in sentences)),       # text slots only (host int: no sync)
            "valid_rows_max": int((attention_mask > 0).sum())}                  # non-pad slots


def batch_pairaudio_pairtext_packed(data, config, tokenizer):
    assert config.dataset_audio_seqlen == config.dataset_text_seqlen
    T = config.dataset_audio_seqlen
    F = config.audiofeat_num_mel_bins * config.audiofeat_stack_length
    buf = PackBuffer(config.dataset_batchsize, T)
    feats, sents = [], []
    for sample in data:
        feat, ids = sample["audiofeat"], sample["input_ids"]
        feat = feat if isinstance(feat, torch.Tensor) else torch.as_tensor(np.asarray(feat))
        tot = feat.shape[0] + len(ids) + 1
        if tot > T:
            continue
        if buf.place(tot):
            yield _emit_asr(buf, feats, sents, F, tokenizer.bos, tokenizer.eos, tokenizer.pad)
            buf.reset()
            feats, sents = [], []
            buf.place(tot)
        feats.append(feat.to(torch.float32))
        sents.append(ids)
    if (not config.dataloader_drop_last_batch) and buf.dirty:
        yield _emit_asr(buf, feats, sents, F, tokenizer.bos, tokenizer.eos, tokenizer.pad)


def _emit_audio(buf: PackBuffer, feats, codes, feat_dim):
    B, T = buf.B, buf.T
    dev = feats[0].device if feats else torch.device("cpu")
    labels = torch.full((B * T,), -100, dtype=torch.int64, device=dev)
    position_ids = np.zeros(B * T, dtype=np.int64)
    attention_mask = np.zeros(B * T, dtype=np.int64)
    sentence_lens = np.ones(B * T, dtype=np.int64)
    input_features = torch.zeros(B, T, feat_dim, dtype=torch.float32, device=dev)
    if len(buf):
        seg, within, flat = buf.scatter_index()
        lens = np.asarray(buf.lens, dtype=np.int64)
        position_ids[flat] = within
        attention_mask[flat] = np.asarray(buf.sents, dtype=np.int64)[seg]
        sentence_lens[flat] = lens[seg]
        # labels: the utterance's codes shifted left by one, the last frame ignored (reference :100-101); the codes
        # stay on the device they were computed on
        for r, c, f, cd in zip(buf.rows, buf.cols, feats, codes):
            n = f.shape[0]
            input_features[r, c:c + n] = f
            if n > 1:
                labels[r * T + c:r * T + c + n - 1] = cd[1:].to(dev)
    lab = labels.view(B, T)
    t = lambda a: torch.from_numpy(a.reshape(B, T))
    return {"input_ids": None, "input_features": input_features, "labels": lab,
            "position_ids": t(position_ids), "attention_mask": t(attention_mask),
            "sentence_lens": t(sentence_lens), "num_sentence": len(buf), "shift_labels": lab}


def batch_audio_packed(data, config, tokenizer):
    """Audio-pretrain packer: a segment = the frames of one utterance; `tokenizer.tokenize(feat)` supplies the
    per-frame codes (device tensor from touchnet_amd.tokenizer.BestRQTokenizer, or any sequence of ints)."""
    T = config.dataset_audio_seqlen
    F = config.audiofeat_num_mel_bins * config.audiofeat_stack_length
    buf = PackBuffer(config.dataset_batchsize, T)
    feats, codes = [], []
    for sample in data:
        feat = sample["audiofeat"]
        feat = feat if isinstance(feat, torch.Tensor) else torch.as_tensor(np.asarray(feat))
        n = feat.shape[0]
        if n > T:
            continue
        if buf.place(n):
            yield _emit_audio(buf, feats, codes, F)
            buf.reset()
            feats, codes = [], []
            buf.place(n)
        feat = feat.to(torch.float32)
        cd = tokenizer.tokenize(feat)
        cd = cd if isinstance(cd, torch.Tensor) else torch.as_tensor(list(cd), dtype=torch.int64)
        assert cd.shape[0] == n
        feats.append(feat)
        codes.append(cd.to(torch.int64))
    if (not config.dataloader_drop_last_batch) and buf.dirty:
        yield _emit_audio(buf, feats, codes, F)


# ---------------------------------------------------------------------------------------------------------------------
# The UNPACKED forms (`--dataset_enable_pack false`): one sample per row, rows right-padded to the longest of the batch,
# dynamic batch size.  processing_touch_audio.py:217-306 (`batch_audio`) and :309-428 (`batch_pairaudio_pairtext`).
# The flush rule is the reference's, including its order of operations: the running maximum is updated BEFORE a sample
# that is too long is skipped (such a sample still widens the budget test until the next flush), a flush happens when
# (samples held + 1) x running maximum exceeds batchsize x seqlen, the new sample opens the next batch.
# ---------------------------------------------------------------------------------------------------------------------
class _DynamicRows:
    def __init__(self, budget: int, limit: int):
        self.budget, self.limit, self.longest, self.rows = int(budget), int(limit), 0, []

    def offer(self, length: int, row):
        """-> the rows to emit now (or None); `row` joins the batch that is open afterwards.  `row = None`: only the
        running maximum moves (a skipped sample)."""
        self.longest = max(self.longest, length)
        if length > self.limit or row is None:
            return None
        out = None
        if (len(self.rows) + 1) * self.longest > self.budget:
            out, self.rows, self.longest = self.rows, [], length
        self.rows.append(row)
        return out


def _pad_rows(seqs, value, dtype):
    n = max(len(s) for s in seqs)
    out = np.full((len(seqs), n), value, dtype=dtype)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = s
    return torch.from_numpy(out)


def _pad_feats(feats):
    n, dim, dev = max(f.shape[0] for f in feats), feats[0].shape[1], feats[0].device
    out = torch.zeros(len(feats), n, dim, dtype=torch.float32, device=dev)     # a device-side frontend keeps features in HBM
    for i, f in enumerate(feats):
        out[i, :f.shape[0]] = f
    return out


def _emit_pairs(rows, pad):
    labels = _pad_rows([r["labels"] for r in rows], -100, np.int64)
    return {"input_ids": _pad_rows([r["input_ids"] for r in rows], pad, np.int64),
            "input_features": _pad_feats([r["feats"] for r in rows]),
            "labels": labels, "shift_labels": labels, "position_ids": None,
            "attention_mask": _pad_rows([np.ones(len(r["labels"]), np.int64) for r in rows], 0, np.int64),
            "sentence_lens": _pad_rows([np.full(len(r["labels"]), r["slen"], np.int64) for r in rows], 1, np.int64),
            "num_sentence": len(rows),
            "labelled_rows_max": int(sum(r["slen"] + 1 for r in rows))}


def batch_pairaudio_pairtext(data, config, tokenizer):
    """processing_touch_audio.py:309-428.  Row = `audio_len` feature frames, then bos + ids; labels ids + eos on the text
    slots; `sentence_lens` = the text length over the whole row."""
    assert config.dataset_audio_seqlen == config.dataset_text_seqlen
    acc = _DynamicRows(config.dataset_batchsize * config.dataset_audio_seqlen, config.dataset_audio_seqlen)
    for sample in data:
        feat, ids = sample["audiofeat"], list(sample["input_ids"])
        alen, total = feat.shape[0], feat.shape[0] + len(ids) + 1
        row = None
        if total <= acc.limit:
            full = torch.zeros(total, feat.shape[1], dtype=torch.float32, device=feat.device)
            full[:alen] = feat
            row = {"feats": full, "slen": len(ids),
                   "input_ids": np.concatenate([np.full(alen, tokenizer.pad), [tokenizer.bos], ids]).astype(np.int64),
                   "labels": np.concatenate([np.full(alen, -100), ids, [tokenizer.eos]]).astype(np.int64)}
        out = acc.offer(total, row)
        if out:
            yield _emit_pairs(out, tokenizer.pad)
    if not config.dataloader_drop_last_batch and acc.rows:
        yield _emit_pairs(acc.rows, tokenizer.pad)


def _emit_audio_rows(rows):
    labels = _pad_rows([r["labels"] for r in rows], -100, np.int64)
    return {"input_ids": None, "input_features": _pad_feats([r["feats"] for r in rows]), "labels": labels,
            "shift_labels": labels, "position_ids": None, "attention_mask": None,
            "sentence_lens": _pad_rows([np.full(len(r["labels"]), len(r["labels"]), np.int64) for r in rows], 1, np.int64),
            "num_sentence": len(rows)}


def batch_audio(data, config, tokenizer):
    """processing_touch_audio.py:217-306 (audio pretraining, unpacked): labels = the BEST-RQ codes shifted by one, the last
    frame unlabelled; no token ids, no mask (the reference leaves both to the model)."""
    acc = _DynamicRows(config.dataset_batchsize * config.dataset_audio_seqlen, config.dataset_audio_seqlen)
    for sample in data:
        feat = sample["audiofeat"]
        n, row = feat.shape[0], None
        if n <= acc.limit:
            codes = list(tokenizer.tokenize(feat))
            assert len(codes) == n
            row = {"feats": feat, "labels": np.asarray(codes[1:] + [-100], dtype=np.int64)}
        out = acc.offer(n, row)
        if out:
            yield _emit_audio_rows(out)
    if not config.dataloader_drop_last_batch and acc.rows:
        yield _emit_audio_rows(acc.rows)
