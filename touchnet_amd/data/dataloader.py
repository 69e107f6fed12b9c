"""`build_dataloader_fn` of the MI355X TrainSpecs — the callable the reference trainer invokes as
`build_dataloader_fn(tokenizer=, data_config=, dp_rank=, dp_world_size=, split=)` (touchnet/bin/train.py:157-170) and
whose result it iterates, checkpoints (`state_dict` / `load_state_dict`) and asks for `get_epoch()`
(touchnet/data/dataloader.py:30-43, 116-180).

MI355X-first: the frontend (PCM -> fbank / log-mel -> stack) runs on the device inside the datapipe stages
(touchnet_amd/data/functions.py), so there is no CPU feature work to spread over worker PROCESSES; what is left on
the host is memory-mapped shard reads and integer packing.  The loader therefore runs the datapipe in ONE background
thread (reads release the GIL), `prefetch` batches ahead, and snapshots the datapipe's resume state WITH every batch,
so that `state_dict()` after k consumed batches resumes at batch k+1 exactly (the reference gets this from torchdata's
StatefulDataLoader worker snapshots).
"""
from __future__ import annotations

import copy
import queue
import threading
from typing import Any, Dict, Optional

from touchnet_amd.data import functions
from touchnet_amd.data.datapipe import LowLevelTouchDatapipe, MidLevelTouchDatapipe


class BaseDataLoader:
    """Interface of touchnet/data/dataloader.py:30-43 (Stateful + `__iter__` + `get_epoch`)."""

    def __iter__(self):
        raise NotImplementedError

    def get_epoch(self) -> int:
        raise NotImplementedError

    def state_dict(self) -> Dict[str, Any]:
        raise NotImplementedError

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        raise NotImplementedError


class PackedDataLoader(BaseDataLoader):
    _END = object()

    def __init__(self, datapipe, dp_rank: int, dp_world_size: int, prefetch: int = 2):
        self.datapipe, self.dp_rank, self.dp_world_size, self.prefetch = datapipe, dp_rank, dp_world_size, prefetch
        self._rank_id = f"dp_rank_{dp_rank}"
        self._state = datapipe.state_dict()          # resume point AFTER the last batch handed out
        self._thread: Optional[threading.Thread] = None
        self._stop = threading.Event()

    # ---- iteration --------------------------------------------------------------------------------------------
    def _produce(self, q: "queue.Queue", stop: threading.Event, device: Optional[int]):
        try:
            stream = None
            if device is not None:
                # The current HIP device is per THREAD and defaults to 0: without this every rank's frontend kernels and
                # batch buffers would land on GPU 0 (extra contexts, memory and a cross-device copy per batch).  The
                # producer also gets its own stream, and every batch carries an event recorded on it: the consumer's
                # stream waits for that event instead of relying on both sides using the default stream.
                import torch
                torch.cuda.set_device(device)
                stream = torch.cuda.Stream(device=device)
            for batch in self._iterate(stream):
                item = (batch, copy.deepcopy(self.datapipe.state_dict()))
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            q.put((self._END, None))
        except BaseException as e:                   # surface producer errors in the consumer (train.py has no recovery)
            q.put((e, None))

    def _iterate(self, stream):
        """The datapipe's batches, produced on `stream` (when there is a device) and tagged with a `_ready` event."""
        if stream is None:
            yield from self.datapipe
            return
        import torch
        it = iter(self.datapipe)
        while True:
            with torch.cuda.stream(stream):
                try:
                    batch = next(it)
                except StopIteration:
                    return
                if isinstance(batch, dict):
                    ev = torch.cuda.Event()
                    ev.record(stream)
                    batch["_ready"] = ev
            yield batch

    def __iter__(self):
        self.shutdown()
        self.datapipe.load_state_dict(copy.deepcopy(self._state))
        q: "queue.Queue" = queue.Queue(maxsize=max(1, self.prefetch))
        self._stop = threading.Event()
        device = None
        try:
            import torch
            if torch.cuda.is_available():
                device = torch.cuda.current_device()       # the CALLER's device (its rank's GPU)
        except ImportError:
            pass
        self._thread = threading.Thread(target=self._produce, args=(q, self._stop, device), daemon=True)
        self._thread.start()
        while True:
            batch, state = q.get()
            if batch is self._END:
                return
            if isinstance(batch, BaseException):
                raise batch
            if isinstance(batch, dict) and "_ready" in batch:
                import torch
                ev = batch.pop("_ready")
                torch.cuda.current_stream().wait_event(ev)  # device-side ordering with the consumer's stream, no host sync
                for v in batch.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(torch.cuda.current_stream())
            self._state = state
            yield batch

    def shutdown(self):
        if self._thread is not None and self._thread.is_alive():
            self._stop.set()
            self._thread.join(timeout=5)
        self._thread = None

    # ---- Stateful (same keys as ParallelAwareDataloader, dataloader.py:81-106) --------------------------------------
    def state_dict(self) -> Dict[str, Any]:
        return {self._rank_id: copy.deepcopy(self._state), "world_size": self.dp_world_size}

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        if not state_dict or self._rank_id not in state_dict:
            return
        assert self.dp_world_size == state_dict["world_size"], (
            "dp_degree is inconsistent before and after checkpoint, dataloader resharding is not supported yet.")
        self._state = copy.deepcopy(state_dict[self._rank_id])

    def get_epoch(self) -> int:
        return int(self._state["epoch"])


def _split_config(data_config, split: str):
    """dataloader.py:122-141: evaluation splits turn shuffling / augmentation off and read their own datalist once."""
    cfg = copy.deepcopy(data_config)
    if split != "train":
        for k in ("datalist_shuffling", "dataset_shuffling", "audio_speed_perturb", "audiofeat_spec_aug",
                  "audiofeat_spec_sub", "audiofeat_spec_trim"):
            setattr(cfg, k, False)
        cfg.audiofeat_dither = 0.0
        path = getattr(cfg, f"datalist_{split}_path", None)
        assert path, f"{split} datalist path is not provided"
        if split == "dev":
            cfg.datalist_sharding = False
        cfg.datalist_epoch = 1
        cfg.datalist_path = path
    return cfg


def build_datapipe(cfg, tokenizer, dp_rank: int, dp_world_size: int):
    """Stage chains of causal_lm_datapipe (processing_llama.py:107-126), touch_audio_datapipe
    (processing_touch_audio.py:431-490) and — packed, which the reference cannot run —
    qwen2_audio_datapipe, with the feature stages on the device."""
    from touchnet_amd.models.llama.processing_llama import batch_text
    from touchnet_amd.models.touch_audio.processing_touch_audio import (batch_audio_packed,
                                                                        batch_pairaudio_pairtext_packed)
    kind = cfg.datapipe_type
    pipe = LowLevelTouchDatapipe(cfg, dp_rank, dp_world_size)
    stage = lambda f, *a: MidLevelTouchDatapipe(pipe, f, *a)
    if kind == "causal_lm":
        pipe = stage(functions.filter_samples, cfg)
        return stage(batch_text, cfg, tokenizer)
    augment = any(getattr(cfg, a, False) for a in ("audiofeat_spec_aug", "audiofeat_spec_sub", "audiofeat_spec_trim"))
    # the device pass (tn_feat_augment) holds at most 16 stripes / substitutions of a kind per utterance; the reference takes
    # any count (touchnet/data/functions.py:193-255) — say so here, not as a bare -22 from the kernel in the middle of an epoch
    for flag, keys in (("audiofeat_spec_aug", ("audiofeat_spec_aug_num_t_mask", "audiofeat_spec_aug_num_f_mask")),
                       ("audiofeat_spec_sub", ("audiofeat_spec_sub_num_t_sub",))):
        for k in keys:
            if getattr(cfg, flag, False) and int(getattr(cfg, k, 0)) > 16:
                raise ValueError(f"{k} = {getattr(cfg, k)}: the MI355X augmentation kernel takes at most 16 per utterance "
                                 f"(the recipes use 1-3)")
    if kind == "touch_audio":
        labels_from_audio = hasattr(tokenizer, "quantizer") or type(tokenizer).__name__ == "BestRQTokenizer"
        if not labels_from_audio:
            pipe = stage(functions.text_tokenize, tokenizer)
        pipe = stage(functions.filter_samples, cfg)
        pipe = stage(functions.audio_resample, cfg)
        if getattr(cfg, "audio_speed_perturb", False):            # wav-level augmentation (processing_touch_audio.py:457-459)
            pipe = stage(functions.audio_speed_perturb, cfg)
        if cfg.audio_feat_type == "fbank":
            pipe = stage(functions.audio_compute_fbank, cfg)
        elif cfg.audio_feat_type == "log_mel_spectrogram":
            pipe = stage(functions.audio_compute_log_mel_spectrogram, cfg)
        else:
            raise NotImplementedError(f"audio_feat_type {cfg.audio_feat_type!r} has no device kernel")
        if augment:                                   # spec_aug -> spec_sub -> spec_trim: one launch per utterance
            pipe = stage(functions.audiofeat_augment, cfg)
        pipe = stage(functions.audiofeat_stack, cfg)
        if labels_from_audio:
            if not cfg.dataset_enable_pack:
                from touchnet_amd.models.touch_audio.processing_touch_audio import batch_audio
                return stage(batch_audio, cfg, tokenizer)
            return stage(batch_audio_packed, cfg, tokenizer)
        if not cfg.dataset_enable_pack:                   # one sample per row, right-padded (processing_touch_audio.py:485-487)
            from touchnet_amd.models.touch_audio.processing_touch_audio import batch_pairaudio_pairtext
            return stage(batch_pairaudio_pairtext, cfg, tokenizer)
        return stage(batch_pairaudio_pairtext_packed, cfg, tokenizer)
    if kind == "qwen2_audio":
        from touchnet_amd.models.qwen2_audio.processing_qwen2_audio import batch_qwen2_audio_packed, dynamic_batch
        if not getattr(cfg, "dataset_enable_pack", True):     # the reference's own form: one sample per row, padded
            return stage(dynamic_batch, cfg, tokenizer)
        return stage(batch_qwen2_audio_packed, cfg, tokenizer)
    if kind == "kimi_audio":
        # kimi_audio_datapipe (processing_kimi_audio.py:227-241): the reference's batcher takes (processor, tokenizer);
        # the features are computed on the device here, the frozen speech tokenizer — when the loader has one — rides on
        # the data config (`speech_tokenizer`: callable(features, mask) -> ids)
        from touchnet_amd.models.kimi_audio.processing_kimi_audio import batch_kimi_audio
        return stage(batch_kimi_audio, cfg, None, tokenizer, getattr(cfg, "speech_tokenizer", None))
    raise NotImplementedError(f"datapipe_type {kind!r}")


def build_dataloader(data_config, tokenizer, dp_rank: int, dp_world_size: int, split: str = "train") -> BaseDataLoader:
    cfg = _split_config(data_config, split)
    return PackedDataLoader(build_datapipe(cfg, tokenizer, dp_rank, dp_world_size), dp_rank, dp_world_size,
                            prefetch=max(1, int(getattr(cfg, "dataloader_prefetch_factor", 2) or 2)))
