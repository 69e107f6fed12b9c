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
    if not w.is_cuda:
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
