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
            "labelled_rows_max": int(sum(len(s) + 1 for s in sentences))}       # text slots only (host int: no sync)


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
