"""Tokenizers on the hot path's caller side.  Only the BEST-RQ label tokenizer lives here: text
tokenizers (HuggingFace) are out of scope and stay the reference's (`touchnet/tokenizer/tokenizer.py:93-233`)."""
from .bestrq import BestRQTokenizer, build_tokenizer  # noqa: F401
