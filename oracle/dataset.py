"""Oracle: TouchDataset .idx/.bin reader and the low-level datapipe's sample order (TEST INFRASTRUCTURE, see
oracle/__init__.py).

Restates, with plain `struct` parsing and whole-file reads (no mmap, no views):
  * IndexReader / TouchDataset.get      touchnet/data/dataset.py:206-306, 487-516 (format :101-109, dtype codes :22-33)
  * LowLevelTouchDatapipe.__iter__       touchnet/data/datapipe.py:54-180
Pinned by tests/golden/touchdataset/ — shards written with the reference's own IndexWriter from its test assets,
whose md5s equal the constants of tests/touchnet/bin/test_make_data.py:25-28 — and tests/golden/touchdataset.npz
(what the reference's reader/datapipe returned on them).
"""
from __future__ import annotations

import json
import struct

import numpy as np
import torch

_CODES = {1: np.uint8, 2: np.int8, 3: np.int16, 4: np.int32, 5: np.int64, 6: np.float64, 7: np.float32, 8: np.uint16}


def read_index(idx_path):
    """-> (dtype, lengths int32[N], byte pointers int64[N], document indices int64[M])"""
    raw = open(idx_path, "rb").read()
    assert raw[:9] == b"MMIDIDX\x00\x00", "bad header"
    version, = struct.unpack_from("<Q", raw, 9)
    assert version == 1, "bad version"
    code, = struct.unpack_from("<B", raw, 17)
    n, m = struct.unpack_from("<QQ", raw, 18)
    o = 34
    lens = np.frombuffer(raw, dtype="<i4", count=n, offset=o)
    ptrs = np.frombuffer(raw, dtype="<i8", count=n, offset=o + 4 * n)
    docs = np.frombuffer(raw, dtype="<i8", count=m, offset=o + 12 * n)
    assert docs[-1] == n
    return _CODES[code], lens, ptrs, docs


def read_item(path_prefix, datatype, idx, offset=0, length=None):
    dtype, lens, ptrs, _ = read_index(f"{path_prefix}/{datatype}.idx")
    size = np.dtype(dtype).itemsize
    if length is None:
        length = int(lens[idx]) - offset
    with open(f"{path_prefix}/{datatype}.bin", "rb") as f:
        f.seek(int(ptrs[idx]) + offset * size)
        return np.frombuffer(f.read(int(length) * size), dtype=dtype)


def iterate(lists, cfg, dp_rank=0, dp_world=1, state=(0, 0, 0)):
    """Yields (state_before, sample dict) in the reference's order.  `lists` = [(dir, datatypes)], audio+metainfo only;
    `cfg` needs the datalist_*/dataset_* fields of touchnet.data.DataConfig used at datapipe.py:54-160."""
    epoch, c_lists, c_samples = state
    while epoch < cfg.datalist_epoch:
        order = list(range(len(lists)))
        if cfg.datalist_shuffling:
            g = torch.Generator()
            g.manual_seed(epoch)
            order = torch.randperm(len(lists), generator=g).tolist()
        if cfg.datalist_sharding:
            order = order[dp_rank::dp_world]
        for li in order[c_lists:]:
            d, kind = lists[li]
            assert kind == "audio+metainfo"
            _, lens, _, _ = read_index(f"{d}/audio.idx")
            n = len(lens)
            g = torch.Generator()
            g.manual_seed(epoch + c_lists)
            idxs = torch.randperm(n, generator=g).tolist() if cfg.dataset_shuffling else list(range(n))
            for si in idxs[c_samples:]:
                meta = json.loads(read_item(d, "metainfo", si).tobytes().decode("utf-8").strip())
                sr, offset, length = meta["sample_rate"], 0, None
                info = meta.get("info")
                if info is not None and cfg.dataset_load_audio_via_segments and info.get("segments") is not None:
                    g = torch.Generator()
                    g.manual_seed(epoch + c_lists + c_samples)
                    seg = info["segments"][torch.randint(len(info["segments"]), (1,), generator=g).item()]
                    offset = int(float(seg["start"]) * sr)
                    length = int(float(seg["end"]) * sr) - offset
                    meta["txt"] = seg["txt"]
                if cfg.dataset_random_cut_audio:
                    total = int(lens[si])
                    lo = cfg.dataset_random_cut_audio_min_length_in_ms / 1000.0 * sr
                    hi = cfg.dataset_random_cut_audio_max_length_in_ms / 1000.0 * sr
                    if total > lo:
                        g = torch.Generator()
                        g.manual_seed(epoch + c_lists + c_samples)
                        length = torch.randint(low=int(lo), high=min(total, int(hi)), size=(1,), generator=g).item()
                        offset = torch.randint(low=0, high=max(1, total - length), size=(1,), generator=g).item()
                meta["pcm"] = read_item(d, "audio", si, offset, length)
                yield (epoch, c_lists, c_samples), meta
                c_samples += 1
            c_samples = 0
            c_lists += 1
        c_samples = c_lists = 0
        epoch += 1
