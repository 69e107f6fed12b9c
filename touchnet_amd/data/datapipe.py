"""Low-level datapipe over TouchDataset shards: same iteration order, sharding, resume state and
sample dictionaries as `LowLevelTouchDatapipe` (touchnet/data/datapipe.py:16-180), plus `MidLevelTouchDatapipe`
(:183-213) so that the reference's stage functions — or the device ones of touchnet_amd/data/functions.py — chain
on it unchanged.

Order is DEFINED by torch's CPU generator (randperm/randint seeded with epoch + consumed counters, :60-63, :90-96,
:137-140, :152-160), so the same calls are made here.  One MI355X-specific option: `config.dataset_keep_pcm16=True`
leaves `waveform` as the int16 samples of the memory-mapped shard ([1, N], zero copy); the device frontend uploads
2 bytes per sample and divides by 32768 in HBM (`tn_pcm16_to_f32`) instead of doing a float32 pass on a CPU worker.
"""
from __future__ import annotations

import json
import warnings
from typing import Any, Dict

import numpy as np
import torch
from torch.utils.data import IterableDataset

from .dataset import TouchDataset


def _perm(n: int, seed: int):
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


class LowLevelTouchDatapipe(IterableDataset):
    def __init__(self, config, dp_rank: int, dp_world_size: int):
        super().__init__()
        self.lists = []
        with open(config.datalist_path, "r") as f:
            for line in f:
                parts = line.strip().split()
                assert len(parts) == 2
                self.lists.append(dict(dir=parts[0], datatypes=parts[1]))
        self.config, self.dp_rank, self.dp_world_size = config, dp_rank, dp_world_size
        self.epoch = self.consumed_lists = self.consumed_samples = 0      # checkpointed state

    def load_state_dict(self, state_dict: Dict[str, Any]):
        self.epoch, self.consumed_lists, self.consumed_samples = (
            state_dict["epoch"], state_dict["consumed_lists"], state_dict["consumed_samples"])

    def state_dict(self) -> Dict[str, Any]:
        return {"epoch": self.epoch, "consumed_lists": self.consumed_lists, "consumed_samples": self.consumed_samples}

    # ------------------------------------------------------------------ one sample
    def _audio_sample(self, ds: TouchDataset, i: int) -> dict:
        cfg = self.config
        meta = json.loads(ds.get(i, "metainfo").tobytes().decode("utf-8").strip())
        offset, length, sr = 0, None, meta["sample_rate"]
        seed = self.epoch + self.consumed_lists + self.consumed_samples
        info = meta.get("info", None)
        if info is not None and cfg.dataset_load_audio_via_segments:       # audio sft: one random segment
            segments = info.get("segments", None)
            if segments is not None:
                g = torch.Generator()
                g.manual_seed(seed)
                seg = segments[torch.randint(len(segments), (1,), generator=g).item()]
                offset = int(float(seg["start"]) * sr)
                length = int(float(seg["end"]) * sr) - offset
                meta["txt"] = seg["txt"]
        if cfg.dataset_random_cut_audio:                                    # audio pretrain: random crop
            total = ds.get_idx(i, "audio")[1]
            lo = cfg.dataset_random_cut_audio_min_length_in_ms / 1000.0 * sr
            hi = cfg.dataset_random_cut_audio_max_length_in_ms / 1000.0 * sr
            assert hi > lo
            if total > lo:
                g = torch.Generator()
                g.manual_seed(seed)
                length = torch.randint(low=int(lo), high=min(total, int(hi)), size=(1,), generator=g).item()
                offset = torch.randint(low=0, high=max(1, total - length), size=(1,), generator=g).item()
        pcm = ds.get(i, "audio", offset=offset, length=length)
        if getattr(cfg, "dataset_keep_pcm16", False):
            with warnings.catch_warnings():          # (a read-only view of the mapping: nobody writes to it)
                warnings.simplefilter("ignore", UserWarning)
                meta["waveform"] = torch.from_numpy(np.ascontiguousarray(pcm)).unsqueeze(0)      # int16 [1, N]
        else:
            meta["waveform"] = torch.from_numpy(pcm.astype(np.float32) / 32768.0).unsqueeze(0)   # [-1, 1) [1, N]
        meta["datatypes"] = "audio+metainfo"
        return meta

    def _sample(self, ds: TouchDataset, kind: str, i: int) -> dict:
        if kind == "metainfo":                                              # text pre-training from raw text
            meta = json.loads(ds.get(i, "metainfo").tobytes().decode("utf-8").strip())
            meta["datatypes"] = "metainfo"
            return meta
        if kind == "texttoken":
            return dict(input_ids=ds.get(i, "texttoken").tolist(), datatypes="texttoken")
        if kind == "audio+metainfo":
            return self._audio_sample(ds, i)
        raise NotImplementedError(f"unsupported datatypes: {kind}")

    # ------------------------------------------------------------------ iteration
    def __iter__(self):
        cfg = self.config
        while self.epoch < cfg.datalist_epoch:
            order = _perm(len(self.lists), self.epoch) if cfg.datalist_shuffling else list(range(len(self.lists)))
            if cfg.datalist_sharding:                                       # shards over dp ranks ...
                assert len(order) >= self.dp_world_size, (
                    f"len(list_idxs) = {len(order)}, it should be equal or larger than dp_world_size = "
                    f"{self.dp_world_size}")
                order = order[self.dp_rank::self.dp_world_size]
            info = torch.utils.data.get_worker_info()                       # ... then over dataloader workers
            wid, nw = (0, 1) if info is None else (info.id, info.num_workers)
            if cfg.datalist_epoch > 1:
                assert len(order) >= nw, f"len(list_idxs) = {len(order)}, it should be equal or larger than num_workers = {nw}"
            order = order[wid::nw]
            for li in order[self.consumed_lists:]:
                kind = self.lists[li]["datatypes"]
                ds = TouchDataset(self.lists[li]["dir"], cfg.dataset_mmap, kind)
                n = len(ds)
                samples = _perm(n, self.epoch + self.consumed_lists) if cfg.dataset_shuffling else list(range(n))
                for si in samples[self.consumed_samples:]:
                    yield self._sample(ds, kind, si)
                    self.consumed_samples += 1
                self.consumed_samples = 0
                self.consumed_lists += 1
            self.consumed_samples = self.consumed_lists = 0
            self.epoch += 1


class MidLevelTouchDatapipe(IterableDataset):
    """`MidLevelTouchDatapipe(source, f, *args)`: iterating applies the generator function `f(iter(source), *args)`."""

    def __init__(self, source, f, *args, **kw):
        assert callable(f)
        self.source, self.f, self.args, self.kw = source, f, args, kw

    def __iter__(self):
        return self.f(iter(self.source), *self.args, **self.kw)

    def apply(self, f):
        return MidLevelTouchDatapipe(self, f, *self.args, **self.kw)

    def load_state_dict(self, state_dict):
        self.source.load_state_dict(state_dict)

    def state_dict(self):
        return self.source.state_dict()
