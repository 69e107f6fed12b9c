"""Datapipe stage functions running the frontend ON THE DEVICE — same generator signature
`f(data_iter, config) -> iterator` as touchnet/data/functions.py, so they compose with the reference's
MidLevelTouchDatapipe(source, f, *args) (touchnet/data/datapipe.py:183-213) unchanged.

    audio_compute_fbank                 functions.py:117-134
    audio_compute_log_mel_spectrogram   functions.py:159-190
    audiofeat_spec_aug / _sub / _trim   functions.py:193-255   (draws on the host, applied in one device pass)
    audiofeat_stack                     functions.py:258-286

Samples carry `waveform` [1, N]: float32 in [-1, 1) (datapipe.py int16 / 32768) or, from
touchnet_amd.data.datapipe with `dataset_keep_pcm16`, the int16 samples themselves (2 bytes per sample over PCIe,
converted in HBM).  It is moved to the current HIP device once and every later stage stays there.
"""
import random

import torch

from touchnet_amd.models.backend import ops


def _dev_wave(sample):
    w = sample["waveform"]
    if not w.is_cuda and torch.cuda.is_available():      # (without a GPU the HIP ops below refuse the tensor loudly)
        w = w.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)
    if w.dtype == torch.int16:
        w = ops().pcm16_to_float(w)
    return w.reshape(-1)


def audio_speed_perturb(data, config):
    """functions.py:99-114: one `random.choice(config.audio_speed_perturb_speeds)` per sample (the reference's call, so the
    global `random` stream stays aligned with the stages behind), then the waveform resampled ON THE DEVICE
    (functional.speed_perturb; the reference hands it to libsox — band-limited interpolation of the same signal, not the
    same bits)."""
    for sample in data:
        speed = random.choice(config.audio_speed_perturb_speeds)
        if speed != 1.0:
            w = ops().speed_perturb(_dev_wave(sample), float(speed))
            sample["waveform"] = w.reshape(1, -1)
        yield sample


def audio_compute_fbank(data, config):
    for sample in data:
        assert sample["sample_rate"] == 16000, "device frontend: 16 kHz only (all reference recipes)"
        assert config.audiofeat_dither == 0.0 and config.audiofeat_frame_length == 25 and \
            config.audiofeat_frame_shift == 10
        sample["audiofeat"] = ops().kaldi_fbank(_dev_wave(sample), config.audiofeat_num_mel_bins)
        yield sample


def audio_compute_log_mel_spectrogram(data, config):
    for sample in data:
        assert sample["sample_rate"] == 16000 and config.audiofeat_n_fft == 400 and config.audiofeat_hop_length == 160
        sample["audiofeat"] = ops().log_mel_spectrogram(_dev_wave(sample), config.audiofeat_num_mel_bins,
                                                        padding=config.audiofeat_padding)
        yield sample


# ---- feature-level augmentation: the reference's draws (its calls of the global `random` module, in its order), applied
# ---- by ONE gather launch per utterance (csrc/frontend.hip::feat_augment_kernel)
def _draw_spec_aug(T: int, F: int, config):
    """functions.py:205-217: (start, length) per time stripe, then per frequency stripe"""
    t_masks, f_masks = [], []
    for _ in range(config.audiofeat_spec_aug_num_t_mask):
        start = random.randint(0, T - 1)
        length = random.randint(1, config.audiofeat_spec_aug_max_t)
        t_masks.append((start, min(T, start + length)))
    for _ in range(config.audiofeat_spec_aug_num_f_mask):
        start = random.randint(0, F - 1)
        length = random.randint(1, config.audiofeat_spec_aug_max_f)
        f_masks.append((start, min(F, start + length)))
    return t_masks, f_masks


def _draw_spec_sub(T: int, config):
    """functions.py:232-239: (start, length, pos <= start) per substitution"""
    subs = []
    for _ in range(config.audiofeat_spec_sub_num_t_sub):
        start = random.randint(0, T - 1)
        length = random.randint(1, config.audiofeat_spec_sub_max_t)
        pos = random.randint(0, start)
        subs.append((start, min(T, start + length), pos))
    return subs


def _draw_spec_trim(T: int, config) -> int:
    """functions.py:250-253: rows kept"""
    length = random.randint(1, config.audiofeat_spec_trim_max_t)
    return T - length if length < T / 2 else T


def _apply(sample, t_masks=(), f_masks=(), subs=(), rows=None):
    x = sample["audiofeat"]
    if not (t_masks or f_masks or subs):
        if rows is not None and rows < x.shape[0]:
            sample["audiofeat"] = x[:rows]                  # (a view of the leading rows; the stack stage copies)
        return sample
    sample["audiofeat"] = ops().feat_augment(x, t_masks, f_masks, subs, rows)
    return sample


def audiofeat_spec_aug(data, config):
    for sample in data:
        T, F = sample["audiofeat"].shape
        t_masks, f_masks = _draw_spec_aug(T, F, config)
        yield _apply(sample, t_masks, f_masks)


def audiofeat_spec_sub(data, config):
    for sample in data:
        yield _apply(sample, subs=_draw_spec_sub(sample["audiofeat"].shape[0], config))


def audiofeat_spec_trim(data, config):
    for sample in data:
        yield _apply(sample, rows=_draw_spec_trim(sample["audiofeat"].shape[0], config))


def audiofeat_augment(data, config):
    """The enabled stages of {spec_aug, spec_sub, spec_trim} in the reference's chain order
    (processing_touch_audio.py:468-473) as ONE launch per utterance.  Chained generators pull one sample through all
    stages before the next, so drawing aug -> sub -> trim per sample here consumes the global `random` stream exactly
    like the reference's three stage functions."""
    for sample in data:
        T, F = sample["audiofeat"].shape
        t_masks, f_masks = _draw_spec_aug(T, F, config) if getattr(config, "audiofeat_spec_aug", False) else ([], [])
        subs = _draw_spec_sub(T, config) if getattr(config, "audiofeat_spec_sub", False) else []
        rows = _draw_spec_trim(T, config) if getattr(config, "audiofeat_spec_trim", False) else None
        yield _apply(sample, t_masks, f_masks, subs, rows)


def audio_resample(data, config):
    """functions.py:83-96.  Every reference recipe stores and trains on 16 kHz audio (`--audio_resample_rate 16000`), where
    the stage is the identity; another rate would need torchaudio's windowed-sinc resampler, which has no kernel here."""
    for sample in data:
        rate = getattr(config, "audio_resample_rate", 16000)
        if "sample_rate" in sample and sample["sample_rate"] != rate:
            raise NotImplementedError(f"audio_resample {sample['sample_rate']} -> {rate}: "
                                      f"resample when the dataset is written (the device frontend is 16 kHz only)")
        yield sample


def audiofeat_stack(data, config):
    for sample in data:
        sample["audiofeat"] = ops().audiofeat_stack(sample["audiofeat"], config.audiofeat_stack_length,
                                                    config.audiofeat_stride_length,
                                                    bool(config.audiofeat_normalize))
        yield sample


# ---- host stages (pure bookkeeping; same generator signatures as touchnet/data/functions.py:32-80) -------------------
def text_tokenize(data, tokenizer):
    """functions.py:32-49: `txt` -> `input_ids` with the reference's tokenizer object (tokenizers themselves are out
    of scope and stay the reference's); bos/eos are added by the packers."""
    for sample in data:
        if "txt" in sample:
            sample["input_ids"] = tokenizer.tokenize(sample["txt"], add_special_tokens=False)
        yield sample


def filter_samples(data, config):
    """functions.py:52-80: drop samples by token count, audio duration (ms) and tokens per 10 ms of audio."""
    for sample in data:
        ntok = len(sample["input_ids"]) if "input_ids" in sample else None
        if ntok is not None and not (config.text_min_length_in_tokens_for_filter <= ntok
                                     <= config.text_max_length_in_tokens_for_filter):
            continue
        if "waveform" in sample:
            ms = sample["waveform"].size(1) / sample["sample_rate"] * 1000.0
            if config.audio_speed_perturb:
                ms *= max(config.audio_speed_perturb_speeds)
            if not (config.audio_min_length_in_ms_for_filter <= ms <= config.audio_max_length_in_ms_for_filter):
                continue
            if ntok is not None and ms > 1e-7:
                ratio = ntok / (ms / 10)
                if not (config.min_text_audio_ratio <= ratio <= config.max_text_audio_ratio):
                    continue
        yield sample
