"""TouchDataset reader: random access into `<dir>/<datatype>.idx` + `.bin` pairs.

Same on-disk format and reader API as touchnet/data/dataset.py (`IndexReader` :206-306, `TouchDataset` :397-516;
format note :101-109, dtype codes :22-33); byte-identical fixtures are pinned by the reference's own md5s
(tests/golden/make_golden.py::touchdataset_case).

    .idx :=  b"MMIDIDX\\0\\0" | u64 version=1 | u8 dtype code | u64 N | u64 M | i32 len[N] | i64 byte_ptr[N] | i64 doc[M]
    .bin :=  the N sequences back to back in that dtype

Both files are memory-mapped once; `get` returns a zero-copy view of the mapping (int16 PCM for "audio"), which is
what the device-side frontend wants: the datapipe can hand the int16 samples to the GPU without a float32 pass
on the host (touchnet_amd/data/datapipe.py).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

_MAGIC = b"MMIDIDX\x00\x00"
# dataset.py:22-33
_DTYPES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float32, 8: np.uint16}
_HEAD = np.dtype([("magic", "S9"), ("version", "<u8"), ("code", "u1"), ("n_seq", "<u8"), ("n_doc", "<u8")])


class IndexReader:
    """Parsed view of one .idx file: `dtype`, `sequence_lengths` (elements), `sequence_pointers` (bytes),
    `document_indices`; `index[i]` -> (byte pointer, length) like the reference's."""

    def __init__(self, idx_path: str):
        self._map = np.memmap(idx_path, mode="r")
        if self._map.size < _HEAD.itemsize:
            raise ValueError(f"bad header, cannot read: {idx_path}")
        head = self._map[:_HEAD.itemsize].view(_HEAD)[0]
        if bytes(head["magic"]).ljust(9, b"\0") != _MAGIC:
            raise ValueError(f"bad header, cannot read: {idx_path}")
        if int(head["version"]) != 1:
            raise ValueError(f"bad version, cannot read: {idx_path}")
        self.dtype = _DTYPES[int(head["code"])]
        self.dtype_size = np.dtype(self.dtype).itemsize
        n, m = int(head["n_seq"]), int(head["n_doc"])
        self.sequence_count, self.document_count = n, m
        o = _HEAD.itemsize
        self.sequence_lengths = self._map[o:o + 4 * n].view("<i4")
        self.sequence_pointers = self._map[o + 4 * n:o + 12 * n].view("<i8")
        self.document_indices = self._map[o + 12 * n:o + 12 * n + 8 * m].view("<i8")
        if not (self.document_indices.size == m and m > 0 and int(self.document_indices[-1]) == n):
            raise ValueError(f"truncated or inconsistent index: {idx_path}")

    def __len__(self) -> int:
        return self.sequence_count

    def __getitem__(self, idx: int) -> Tuple[np.int64, np.int32]:
        return self.sequence_pointers[idx], self.sequence_lengths[idx]


class TouchDataset:
    """`TouchDataset(path_prefix, mmap=True, datatypes="audio+metainfo")`: `len()`, `get_idx(i, datatype)`,
    `get(i, datatype, offset=0, length=None)` -> numpy array of that datatype's dtype (a view of the mapping)."""

    def __init__(self, path_prefix: str, mmap: bool = True, datatypes: str = "audio+metainfo"):
        self.path_prefix, self.mmap, self.datatypes = path_prefix, mmap, datatypes
        self.index, self._bin = {}, {}
        for d in datatypes.split("+"):
            idx_path, bin_path = f"{path_prefix}/{d}.idx", f"{path_prefix}/{d}.bin"
            if not (os.path.exists(idx_path) and os.path.exists(bin_path)):
                raise FileNotFoundError(f"One or both of the .idx and .bin files cannot be found at the path prefix "
                                        f"{path_prefix}")
            self.index[d] = IndexReader(idx_path)
            # mmap=False (the reference's FileBinReader) reads through the page cache as well; one code path
            self._bin[d] = np.memmap(bin_path, mode="r") if os.path.getsize(bin_path) else np.zeros(0, np.uint8)
        n = {len(ix) for ix in self.index.values()}
        if len(n) != 1:
            raise ValueError(f"datatypes of {path_prefix} disagree on the number of sequences: {n}")

    def __len__(self) -> int:
        return len(next(iter(self.index.values())))

    def get_idx(self, idx: int, datatype: str):
        return self.index[datatype][idx]

    def get(self, idx: int, datatype: str, offset: int = 0, length: Optional[int] = None) -> np.ndarray:
        ix = self.index[datatype]
        ptr, n = ix[idx]
        if length is None:
            length = int(n) - offset
        start = int(ptr) + offset * ix.dtype_size
        return self._bin[datatype][start:start + int(length) * ix.dtype_size].view(ix.dtype)

    def __getstate__(self):
        return self.path_prefix, self.mmap, self.datatypes

    def __setstate__(self, state):
        self.__init__(*state)
