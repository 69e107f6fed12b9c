"""Datapipe stage functions running the frontend ON THE DEVICE — same generator signature
`f(data_iter, config) -> iterator` as touchnet/data/functions.py, so they compose with the reference's
MidLevelTouchDatapipe(source, f, *args) (touchnet/data/datapipe.py:183-213) unchanged.

    audio_compute_fbank                 functions.py:117-134
    audio_compute_log_mel_spectrogram   functions.py:159-190
    audiofeat_stack                     functions.py:258-286

Samples carry `waveform` [1, N]: float32 in [-1, 1) (datapipe.py int16 / 32768) or, from
touchnet_amd.data.datapipe with `dataset_keep_pcm16`, the int16 samples themselves (2 bytes per sample over PCIe,
converted in HBM).  It is moved to the current HIP device once and every later stage stays there.
"""
import torch

from touchnet_amd.models.backend import ops


def _dev_wave(sample):
    w = sample["waveform"]
    if not w.is_cuda and torch.cuda.is_available():      # (without a GPU the HIP ops below refuse the tensor loudly)
        w = w.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)
    if w.dtype == torch.int16:
        w = ops().pcm16_to_float(w)
    return w.reshape(-1)


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
