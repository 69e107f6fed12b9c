"""BEST-RQ random-projection tokenizer on the MI355X — drop-in for `BestRQTokenizer`
(touchnet/tokenizer/tokenizer.py:236-318; built by `build_tokenizer`, :321-334, from the same `TokenizerConfig` fields,
touchnet/tokenizer/__init__.py:29-60).

The frozen projection and codebook are DEFINED by torch's CPU generator stream (xavier-uniform then normal draws from
one generator seeded with `tokenizer_bestrq_init_seed`, tokenizer.py:256-260), so they are drawn on the host with
the same calls and uploaded once; `tokenize` runs `tn_bestrq_tokenize` on the device and returns a device int64
tensor (the reference returns a Python list; `batch_audio_packed` here consumes the tensor directly, and
`.tolist()` gives the reference's value).  CPU inputs are refused: there is no host fallback.
"""
from __future__ import annotations

import torch

from touchnet_amd import functional as F


class BestRQTokenizer:
    def __init__(self, config, device="cuda", **kwargs):
        self.name = f"BestRQ-{config.tokenizer_bestrq_init_method}-init"
        self.config, self.device = config, torch.device(device)
        self._quantizer = self._codebook = None

    def _build_quantizer_and_codebook(self):
        if self._quantizer is not None:
            return
        c = self.config
        if c.tokenizer_bestrq_init_method != "default":
            raise NotImplementedError(f"Initialization method {c.tokenizer_bestrq_init_method} is not implemented.")
        g = torch.Generator().manual_seed(c.tokenizer_bestrq_init_seed)
        q = torch.empty(c.tokenizer_bestrq_input_size, c.tokenizer_bestrq_emb_size)
        cb = torch.empty(c.tokenizer_bestrq_vocab_size, c.tokenizer_bestrq_emb_size)
        torch.nn.init.xavier_uniform_(q, generator=g)          # best-rq, arXiv:2202.01855
        torch.nn.init.normal_(cb, generator=g)
        cb = torch.nn.functional.normalize(cb, dim=1, p=2, eps=1e-8)
        self._quantizer, self._codebook = q.to(self.device), cb.contiguous().to(self.device)

    @property
    def vocab_size(self):
        return self.config.tokenizer_bestrq_vocab_size

    @property
    def vocab(self):
        return None

    @property
    def inv_vocab(self):
        self._build_quantizer_and_codebook()
        return self._codebook

    decoder = inv_vocab

    def tokenize(self, inputs: torch.Tensor, **kwargs) -> torch.Tensor:
        """inputs [T, input_size] fp32 on the device -> int64 codes [T] on the device."""
        self._build_quantizer_and_codebook()
        return F.bestrq_tokenize(inputs, self._quantizer, self._codebook)

    def detokenize(self, token_ids, **kwargs):
        self._build_quantizer_and_codebook()
        return torch.index_select(self._codebook, 0, token_ids.to(self._codebook.device))

    eos = bos = pad = None


def build_tokenizer(args, **kwargs):
    """tokenizer.py:321-334 for the one tokenizer type in scope."""
    if args.tokenizer_type == "BestRQTokenizer":
        return BestRQTokenizer(args, **kwargs)
    raise NotImplementedError(f"{args.tokenizer_type}: only BestRQTokenizer runs on the device; text tokenizers stay "
                              "the reference's HuggingFaceTokenizer")
