"""TouchDataset WRITER: `<dir>/<datatype>.bin` + `.idx` pairs in the reference's on-disk format (the inverse of
touchnet_amd/data/dataset.py; format note there).  Role of touchnet/bin/make_data.py's `DataBuilder` (:24-96) over
touchnet/data/dataset.py's `IndexWriter` (:94-196) — same method names (`add_item`, `add_document`, `end_document`,
`finalize`), same bytes: the files it writes for the reference's two test utterances have the md5s the reference's own
test pins (tests/touchnet/bin/test_make_data.py:25-28; tests/test_dataset.py).

What is NOT here is make_data.py's command line around it (ffmpeg decoding of arbitrary audio, multiprocessing over a
jsonl): `write_audio_shards` takes PCM that is already 16-bit mono at the dataset's rate.
"""
from __future__ import annotations

import json
import os
import struct
from typing import Iterable, List, Sequence, Tuple

import numpy as np

from touchnet_amd.data.dataset import _DTYPES, _MAGIC

_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


class DataBuilder:
    def __init__(self, bin_path: str, dtype=np.int32):
        self._bin = open(bin_path, "wb")
        self.dtype = np.dtype(dtype)
        if self.dtype not in _CODES:
            raise ValueError(f"dtype {self.dtype} has no code in the index format")
        self.sequence_lengths: List[int] = []
        self.document_indices: List[int] = [0]

    def add_item(self, item) -> None:
        """one sequence (array-like or a tensor); close its document with `end_document`"""
        a = np.asarray(item.numpy() if hasattr(item, "numpy") else item).astype(self.dtype, copy=False)
        self._bin.write(np.ascontiguousarray(a).tobytes())
        self.sequence_lengths.append(int(a.size))

    def add_document(self, items, lengths: Sequence[int]) -> None:
        """a whole document: its sequences back to back in `items`, their `lengths`"""
        a = np.asarray(items.numpy() if hasattr(items, "numpy") else items).astype(self.dtype, copy=False)
        self._bin.write(np.ascontiguousarray(a).tobytes())
        self.sequence_lengths.extend(int(n) for n in lengths)
        self.document_indices.append(len(self.sequence_lengths))

    def end_document(self) -> None:
        self.document_indices.append(len(self.sequence_lengths))

    def finalize(self, idx_path: str) -> None:
        self._bin.close()
        lens = np.asarray(self.sequence_lengths, dtype="<i4")
        ptrs = np.zeros(len(lens), dtype="<i8")
        if len(lens) > 1:
            np.cumsum(lens[:-1].astype(np.int64) * self.dtype.itemsize, out=ptrs[1:])
        docs = np.asarray(self.document_indices, dtype="<i8")
        with open(idx_path, "wb") as f:
            f.write(_MAGIC + struct.pack("<Q", 1) + struct.pack("<B", _CODES[self.dtype]))
            f.write(struct.pack("<Q", len(lens)) + struct.pack("<Q", len(docs)))
            f.write(lens.tobytes() + ptrs.tobytes() + docs.tobytes())


def write_audio_shards(samples: Iterable[Tuple[dict, np.ndarray]], save_dir: str, samples_per_shard: int,
                       sample_rate: int = 16000) -> List[str]:
    """`(metainfo dict, int16 PCM)` pairs -> shard directories `save_dir/000000000`, ... each holding audio.{bin,idx}
    (int16) and metainfo.{bin,idx} (the JSON line of the sample with `sample_rate` appended, UTF-8 bytes): the layout
    make_data.py produces for `--datatypes audio+metainfo` (:158-236)."""
    samples = list(samples)
    shards = []
    for first in range(0, len(samples), samples_per_shard):
        d = "{}/{:09d}".format(save_dir, first // samples_per_shard)
        os.makedirs(d, exist_ok=True)
        audio, meta = DataBuilder(f"{d}/audio.bin", np.int16), DataBuilder(f"{d}/metainfo.bin", np.uint8)
        for info, pcm in samples[first:first + samples_per_shard]:
            info = dict(info)
            info["sample_rate"] = sample_rate
            audio.add_item(np.asarray(pcm, dtype=np.int16))
            audio.end_document()
            meta.add_item(np.frombuffer(json.dumps(info, ensure_ascii=False).strip().encode("utf-8"), dtype=np.uint8))
            meta.end_document()
        audio.finalize(f"{d}/audio.idx")
        meta.finalize(f"{d}/metainfo.idx")
        shards.append(d)
    return shards
