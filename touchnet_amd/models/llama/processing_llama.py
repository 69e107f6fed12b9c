"""Sequence packer for text: the datapipe stage `batch_text(data, config, tokenizer)` with the
reference's generator signature and bit-identical outputs
(touchnet/models/llama/processing_llama.py:24-104), built differently: placement is decided on plain
integers, the five [B, T] int64 buffers are then filled with a handful of vectorised numpy scatters
instead of per-sentence tensor slice assignments.

Batch dict contract: input_ids (bos + ids, pad elsewhere), labels (ids + eos,
PRE-shifted, -100 elsewhere), position_ids (restart per sentence), attention_mask (document index
1.. per row, 0 = pad), sentence_lens (len+1 on the sentence, 1 on pad), num_sentence (python int),
inputs_embeds = None.
"""
from __future__ import annotations

import numpy as np
import torch


class PackBuffer:
    """Greedy first-fit placement of variable-length segments into B rows of T slots."""

    def __init__(self, batchsize: int, seqlen: int):
        self.B, self.T = batchsize, seqlen
        self.reset()

    def reset(self):
        self.row, self.col, self.sent = 0, 0, 1
        self.rows, self.cols, self.lens, self.sents = [], [], [], []

    def __len__(self):
        return len(self.lens)

    def place(self, n: int) -> bool:
        """Reserve n slots.  Returns True when the buffer had to be flushed FIRST (caller emits, resets
        and calls again) — i.e. the segment does not fit the last row."""
        if n > self.T:
            # the reference fails here too (slice-size mismatch at processing_llama.py:86-90); say why
            raise ValueError(f"segment of {n} slots does not fit a row of {self.T}: filter or truncate the sample "
                             f"before packing (text_max_length_in_tokens_for_filter)")
        if self.col + n > self.T:
            if self.row == self.B - 1:
                return True
            self.row, self.col, self.sent = self.row + 1, 0, 1
        self.rows.append(self.row)
        self.cols.append(self.col)
        self.lens.append(n)
        self.sents.append(self.sent)
        self.col += n
        self.sent += 1
        return False

    @property
    def dirty(self) -> bool:
        return self.col > 0 or self.row > 0

    def scatter_index(self):
        lens = np.asarray(self.lens, dtype=np.int64)
        seg = np.repeat(np.arange(lens.size), lens)
        within = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
        flat = (np.asarray(self.rows, dtype=np.int64)[seg] * self.T
                + np.asarray(self.cols, dtype=np.int64)[seg] + within)
        return seg, within, flat


def _emit_text(buf: PackBuffer, sentences, bos, eos, pad):
    B, T = buf.B, buf.T
    input_ids = np.full(B * T, pad, dtype=np.int64)
    labels = np.full(B * T, -100, dtype=np.int64)
    position_ids = np.zeros(B * T, dtype=np.int64)
    attention_mask = np.zeros(B * T, dtype=np.int64)
    sentence_lens = np.ones(B * T, dtype=np.int64)
    if len(buf):
        seg, within, flat = buf.scatter_index()
        lens = np.asarray(buf.lens, dtype=np.int64)
        toks = np.concatenate([np.asarray(s, dtype=np.int64) for s in sentences]) if sentences else np.zeros(0, np.int64)
        starts = np.cumsum(lens) - lens
        # inputs: bos at the head of every segment, tokens after it; labels: tokens, eos at the tail
        inp = np.empty(flat.size, dtype=np.int64)
        lab = np.empty(flat.size, dtype=np.int64)
        is_head = within == 0
        is_tail = within == lens[seg] - 1
        tok_pos_in = np.nonzero(~is_head)[0]
        tok_pos_lab = np.nonzero(~is_tail)[0]
        inp[is_head] = bos
        inp[tok_pos_in] = toks
        lab[is_tail] = eos
        lab[tok_pos_lab] = toks
        input_ids[flat], labels[flat] = inp, lab
        position_ids[flat] = within
        attention_mask[flat] = np.asarray(buf.sents, dtype=np.int64)[seg]
        sentence_lens[flat] = lens[seg]
        del starts
    t = lambda a: torch.from_numpy(a.reshape(B, T))
    return {"input_ids": t(input_ids), "inputs_embeds": None, "labels": t(labels), "position_ids": t(position_ids),
            "attention_mask": t(attention_mask), "sentence_lens": t(sentence_lens), "num_sentence": len(buf),
            "labelled_rows_max": int(sum(buf.lens)),          # every non-pad slot carries a label (host int: no sync)
            # non-pad slots of the batch: the decoder drops the padding slots from its row-wise work (DecoderModel.forward)
            "valid_rows_max": int(sum(buf.lens))}


def batch_text(data, config, tokenizer):
    """Datapipe stage: iterator of {'input_ids': list[int]} -> iterator of packed batch dicts."""
    buf = PackBuffer(config.dataset_batchsize, config.dataset_text_seqlen)
    pending = []
    for sample in data:
        ids = sample["input_ids"]
        n = len(ids) + 1                      # +1 for bos / eos
        if buf.place(n):
            yield _emit_text(buf, pending, tokenizer.bos, tokenizer.eos, tokenizer.pad)
            buf.reset()
            pending = []
            buf.place(n)
        pending.append(ids)
    if (not config.dataloader_drop_last_batch) and buf.dirty:
        yield _emit_text(buf, pending, tokenizer.bos, tokenizer.eos, tokenizer.pad)
