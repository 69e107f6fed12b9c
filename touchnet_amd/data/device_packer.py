"""Device-side sequence packers: same datapipe-stage signatures and BIT-IDENTICAL batches as the host packers
(`batch_text`, `batch_pairaudio_pairtext_packed`; reference: touchnet/models/llama/processing_llama.py:24-104,
touchnet/models/touch_audio/processing_touch_audio.py:117-214), but placement and scatter run on the MI355X
(csrc/packer.hip: tn_pack_plan + tn_pack_fill) from a length list, and the buffers are born in HBM.

Samples are taken in WINDOWS (enough segments for a few batches): one upload of the window's lengths + tokens, one
placement launch, one fill launch per batch, ONE host read-back per window (how many batches the window closed).
A window's last batch is open (the next sample might still fit): its segments are carried into the next window, so
the stream of batches equals the one-sample-at-a-time packers' exactly; the tail batch is emitted unless
`dataloader_drop_last_batch`.
"""
from __future__ import annotations

from typing import Iterator, List, Optional

import torch

from touchnet_amd import _C


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _i32(vals, dev):
    return torch.tensor(vals, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)


def _i64(vals, dev):
    return torch.tensor(vals, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)


class _Window:
    """Segments waiting to be packed: token lists, optional device feature tensors."""

    def __init__(self):
        self.tokens: List[List[int]] = []
        self.feats: List[Optional[torch.Tensor]] = []

    def __len__(self):
        return len(self.tokens)

    def add(self, ids, feat=None):
        self.tokens.append(ids)
        self.feats.append(feat)

    def tail(self, keep_from: int) -> "_Window":
        w = _Window()
        w.tokens, w.feats = self.tokens[keep_from:], self.feats[keep_from:]
        return w


def _pack_window(win: _Window, B: int, T: int, F: int, bos: int, eos: int, pad: int, audio: bool, final: bool,
                 drop_last: bool):
    """-> (list of batch dicts, index of the first segment NOT emitted)."""
    lib, p, st = _C.lib(), _C.ptr, _C.stream
    dev = _dev()
    n = len(win)
    ntok = [len(t) for t in win.tokens]
    alen = [int(f.shape[0]) if f is not None else 0 for f in win.feats]
    lens = [a + t + 1 for a, t in zip(alen, ntok)]
    d_lens, d_ntok = _i32(lens, dev), _i32(ntok, dev)
    tok_off, feat_off, o, fo = [], [], 0, 0
    for t, a in zip(ntok, alen):
        tok_off.append(o)
        feat_off.append(fo)
        o += t
        fo += a
    flat = [v for t in win.tokens for v in t] or [0]
    d_tokens, d_tok_off = _i64(flat, dev), _i64(tok_off, dev)
    d_alen = _i32(alen, dev) if audio else None
    d_feat_off = _i64(feat_off, dev) if audio else None
    d_feat = (torch.cat([f.to(dev, torch.float32) for f in win.feats if f is not None and f.shape[0]], dim=0)
              if audio and fo > 0 else None)
    row, col, sent, batch = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    counts = torch.zeros(3, dtype=torch.int32, device=dev)
    _C.check(lib.tn_pack_plan(p(d_lens), n, B, T, int(audio), p(row), p(col), p(sent), p(batch), p(counts), st()),
             "tn_pack_plan")
    nb, too_long, err = [int(v) for v in counts.tolist()]           # the window's only host read-back
    if err:
        raise ValueError(f"a segment does not fit a row of {T}: filter or truncate the sample before packing "
                         f"(text_max_length_in_tokens_for_filter)")
    # the last batch of a window stays open unless the stream ends here
    emit = nb if (final and not drop_last) else max(0, nb - 1)
    out = []
    for b in range(emit):
        bufs = [torch.empty(B, T, dtype=torch.int64, device=dev) for _ in range(5)]
        feats = torch.empty(B, T, F, dtype=torch.float32, device=dev) if audio else None
        nsent = torch.zeros(1, dtype=torch.int32, device=dev)
        _C.check(lib.tn_pack_fill(p(row), p(col), p(sent), p(batch), b, n, p(d_ntok), p(d_tok_off), p(d_tokens),
                                  p(d_alen), p(d_feat_off), p(d_feat), F, B, T, bos, eos, pad, *[p(x) for x in bufs],
                                  p(feats), p(nsent), st()), "tn_pack_fill")
        d = {"input_ids": bufs[0], "labels": bufs[1], "position_ids": bufs[2], "attention_mask": bufs[3],
             "sentence_lens": bufs[4], "num_sentence": nsent.to(torch.float32)}    # device scalar: no host sync
        if audio:
            d["input_features"] = feats
            d["shift_labels"] = bufs[1]
        else:
            d["inputs_embeds"] = None
        out.append(d)
    if final:
        return out, n
    # first segment of the open batch (batch indices are non-decreasing along the window; skipped ones are -1)
    bt = batch.tolist() if nb > 0 else []
    first_open = next((i for i, v in enumerate(bt) if v == nb - 1), n)
    return out, first_open


def _stream(samples, B, T, F, tok, audio, drop_last, window):
    win = _Window()
    for ids, feat in samples:
        win.add(ids, feat)
        if len(win) >= window:
            batches, keep = _pack_window(win, B, T, F, tok.bos, tok.eos, tok.pad, audio, False, drop_last)
            yield from batches
            win = win.tail(keep)
            if len(win) >= window:       # one open batch larger than the window: grow it
                window *= 2
    if len(win):
        batches, _ = _pack_window(win, B, T, F, tok.bos, tok.eos, tok.pad, audio, True, drop_last)
        yield from batches


def batch_text_device(data: Iterator[dict], config, tokenizer, window: int = 4096):
    """Datapipe stage == batch_text (processing_llama.py:24-104), buffers packed on the device."""
    return _stream(((s["input_ids"], None) for s in data), config.dataset_batchsize, config.dataset_text_seqlen, 0,
                   tokenizer, False, config.dataloader_drop_last_batch, window)


def batch_pairaudio_pairtext_packed_device(data: Iterator[dict], config, tokenizer, window: int = 1024):
    """Datapipe stage == batch_pairaudio_pairtext_packed (processing_touch_audio.py:117-214): features come from the
    device frontend and never leave HBM."""
    assert config.dataset_audio_seqlen == config.dataset_text_seqlen
    F = config.audiofeat_num_mel_bins * config.audiofeat_stack_length
    return _stream(((s["input_ids"], s["audiofeat"]) for s in data), config.dataset_batchsize,
                   config.dataset_audio_seqlen, F, tokenizer, True, config.dataloader_drop_last_batch, window)
