"""Oracle: audio frontend (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, in NumPy float32/float64 on the CPU,
  * ``audiofeat_stack``                     touchnet/data/functions.py:258-286
  * ``spec_aug / spec_sub / spec_trim``     touchnet/data/functions.py:193-255
  * ``speed_perturb``                       touchnet/data/functions.py:99-114 (sox: PARITY UNPINNED, see the function)
  * ``audio_compute_log_mel_spectrogram``   touchnet/data/functions.py:159-190
    (librosa.filters.mel(sr, n_fft, n_mels) = slaney scale + slaney norm is a
    third-party dependency, pyproject.toml:17 `librosa>=0.11.0`; restated from its
    published definition and pinned against transformers.audio_utils.mel_filter_bank
    and the reference run on tests/assets/dataset/*.wav, see tests/golden/)
  * ``audio_compute_fbank``                 touchnet/data/functions.py:117-134
    -> torchaudio.compliance.kaldi.fbank (pyproject.toml `torchaudio>=2.7.0`, NOT
    installed here, no golden vector in the reference): follows the published Kaldi
    compute-fbank-feats algorithm with torchaudio's defaults and the reference's overrides
    (energy_floor=0.0, dither=cfg(0.0), input * 32768).  NOT pinned against torchaudio
    itself; pinned against an independent third-party implementation of the same function,
    transformers.audio_utils' Kaldi-compatible fbank, on the reference's two test wavs
    (tests/golden/kaldi_fbank_hf.npz, agreement 1.5e-4 on values in [-16, 24]).
"""
from __future__ import annotations

import math

import numpy as np


# --------------------------------------------------------------------------- stack
def audiofeat_stack(feat, stack, stride, normalize=True):
    """functions.py:258-286.  feat float32 [T, D] -> [ceil(T/stride), D*stack]."""
    x = np.asarray(feat, dtype=np.float32)
    T0, D = x.shape
    t_lfr = int(math.ceil(T0 / stride))
    lp = (stack - 1) // 2
    x = np.concatenate([np.repeat(x[:1], lp, axis=0), x], axis=0)
    T = T0 + lp
    last_idx = (T - stack) // stride + 1
    num_padding = stack - (T - last_idx * stride)
    if num_padding > 0:
        # the reference's closed form (functions.py:277-279), float division included
        num_padding = (2 * stack - 2 * T + (t_lfr - 1 + last_idx) * stride) / 2 * (t_lfr - last_idx)
        x = np.concatenate([x] + [x[-1:]] * int(num_padding), axis=0)
    need = (t_lfr - 1) * stride + stack
    if need > x.shape[0]:
        raise ValueError("as_strided window runs past the padded buffer (reference would fault)")
    out = np.stack([x[i * stride:i * stride + stack].reshape(-1) for i in range(t_lfr)], axis=0)
    if normalize:
        mean = out.mean(axis=-1, keepdims=True, dtype=np.float32)
        std = out.std(axis=-1, keepdims=True, ddof=1, dtype=np.float32)
        out = (out - mean) / (std + np.float32(1e-5))
    return out.astype(np.float32)


# --------------------------------------------------------------------------- speed perturbation
def speed_perturb(x, speed, zeros=32, rolloff=0.95, beta=14.769656459379492):
    """touchnet/data/functions.py:99-114 (sox `speed s` + `rate sr`): the waveform resampled by 1 / s.  PARITY UNPINNED: libsox
    is a third-party binary resampler, absent here like torchaudio; this is the float64 evaluation of the band-limited
    interpolation the product kernel implements, written directly from the formula (no table):
        y[n] = sum_k x[k] h(n s - k),  h(t) = c sinc(c t) kaiser(t / W),  c = rolloff min(1, 1 / s),  W = zeros / c
    with floor(N / s) output samples."""
    import fractions
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    if speed == 1.0:
        return x.astype(np.float32)
    fr = fractions.Fraction(str(speed)).limit_denominator(1000)
    s = fr.numerator / fr.denominator
    n_out = max((x.size * fr.denominator) // fr.numerator, 1)
    c = rolloff * min(1.0, 1.0 / s)
    W = zeros / c
    half = int(math.ceil(W)) + 1
    y = np.zeros(n_out, dtype=np.float64)
    i0b = np.i0(beta)
    for lo in range(0, n_out, 8192):                       # (blocks of outputs: the same sum, evaluated as arrays)
        n = np.arange(lo, min(lo + 8192, n_out), dtype=np.int64)
        pos = n * fr.numerator / fr.denominator
        k = np.floor(pos).astype(np.int64)[:, None] + np.arange(-half, half + 1, dtype=np.int64)[None, :]
        t = pos[:, None] - k
        w = np.where(np.abs(t) < W, np.i0(beta * np.sqrt(np.clip(1.0 - (t / W) ** 2, 0.0, None))) / i0b, 0.0)
        inside = (k >= 0) & (k < x.size)
        y[lo:lo + n.size] = (np.where(inside, x[np.clip(k, 0, x.size - 1)], 0.0) * (c * np.sinc(c * t) * w)).sum(1)
    return y.astype(np.float32)


# --------------------------------------------------------------------------- feature-level augmentation
# touchnet/data/functions.py:193-255, loop by loop, with the reference's calls of the global `random` module in the
# reference's order (pinned: tests/golden/audiofeat_augment.npz holds outputs of the reference's own stage functions
# for given seeds).
def spec_aug(x, num_t_mask, num_f_mask, max_t, max_f, rng):
    """functions.py:193-217.  `rng`: the `random` module (or a random.Random)."""
    y = np.array(x, dtype=np.float32, copy=True)
    max_frames, max_freq = y.shape
    for _ in range(num_t_mask):
        start = rng.randint(0, max_frames - 1)
        length = rng.randint(1, max_t)
        y[start:min(max_frames, start + length), :] = 0
    for _ in range(num_f_mask):
        start = rng.randint(0, max_freq - 1)
        length = rng.randint(1, max_f)
        y[:, start:min(max_freq, start + length)] = 0
    return y


def spec_sub(x, num_t_sub, max_t, rng):
    """functions.py:220-239: rows [start, end) <- rows [start - pos, end - pos) of the INPUT."""
    x = np.asarray(x, dtype=np.float32)
    y = x.copy()
    max_frames = y.shape[0]
    for _ in range(num_t_sub):
        start = rng.randint(0, max_frames - 1)
        length = rng.randint(1, max_t)
        end = min(max_frames, start + length)
        pos = rng.randint(0, start)
        y[start:end, :] = x[start - pos:end - pos, :]
    return y


def spec_trim(x, max_t, rng):
    """functions.py:242-255: drop `length` tail frames when that is less than half of them."""
    x = np.asarray(x, dtype=np.float32)
    max_frames = x.shape[0]
    length = rng.randint(1, max_t)
    return x[:max_frames - length].copy() if length < max_frames / 2 else x


# --------------------------------------------------------------------------- log-mel
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filters(sr, n_fft, n_mels):
    """librosa.filters.mel(sr=, n_fft=, n_mels=) defaults: fmin 0, fmax sr/2, htk=False,
    norm='slaney'.  Returns float32 [n_mels, 1 + n_fft//2]."""
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, fftfreqs.size), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def log_mel_spectrogram(wav, sr=16000, n_fft=400, hop=160, n_mels=128, padding=0):
    """functions.py:159-190: hann(periodic) STFT, center=True reflect pad, |.|^2 with the
    last frame dropped, slaney mel, log10(clamp 1e-10), max(x, max-8), (x+4)/4.
    wav float32 [N] -> float32 [N//hop, n_mels]."""
    x = np.asarray(wav, dtype=np.float32)
    if padding > 0:
        x = np.concatenate([x, np.zeros(padding, dtype=np.float32)])
    n = np.arange(n_fft, dtype=np.float64)
    window = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)
    xp = np.pad(x, (n_fft // 2, n_fft // 2), mode="reflect")
    n_frames = 1 + (xp.size - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = xp[idx] * window[None, :]
    spec = np.fft.rfft(frames.astype(np.float32), axis=-1)           # [frames, 201]
    mag = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2)[:-1]
    mel = mag @ slaney_mel_filters(sr, n_fft, n_mels).T              # [frames-1, n_mels]
    log_spec = np.log10(np.maximum(mel, np.float32(1e-10)))
    log_spec = np.maximum(log_spec, log_spec.max() - np.float32(8.0))
    return ((log_spec + np.float32(4.0)) / np.float32(4.0)).astype(np.float32)


# --------------------------------------------------------------------------- kaldi fbank
def kaldi_mel_banks(num_bins, padded, sr, low_freq=20.0, high_freq=0.0):
    """Kaldi MelBanks (no VTLN): triangles equally spaced on mel = 1127 ln(1 + f/700)."""
    nyq = 0.5 * sr
    if high_freq <= 0.0:
        high_freq += nyq
    n_fft_bins = padded // 2
    bin_w = sr / padded
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    mel_lo, mel_hi = mel(low_freq), mel(high_freq)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_lo + b * delta, mel_lo + (b + 1) * delta, mel_lo + (b + 2) * delta
    m = mel(bin_w * np.arange(n_fft_bins, dtype=np.float64))[None, :]
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    banks = np.maximum(0.0, np.minimum(up, down))
    return np.pad(banks, ((0, 0), (0, 1))).astype(np.float32)          # [bins, padded/2+1]


def povey_window(n):
    k = np.arange(n, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1))) ** 0.85).astype(np.float32)


def kaldi_fbank(wav, sr=16000, num_mel_bins=80, frame_length_ms=25.0, frame_shift_ms=10.0,
                dither=0.0, preemph=0.97):
    """functions.py:117-134 (waveform * 32768, then torchaudio.compliance.kaldi.fbank with
    energy_floor=0.0, snip_edges, remove_dc_offset, povey window, round_to_power_of_two,
    use_power, use_log_fbank, low_freq=20, high_freq=nyquist).  wav float32 [N] in [-1,1)
    -> float32 [1 + (N - win)//shift, num_mel_bins]."""
    assert dither == 0.0, "all reference recipes use dither 0.0 (random otherwise)"
    x = np.asarray(wav, dtype=np.float32) * np.float32(1 << 15)
    win = int(sr * frame_length_ms * 0.001)
    shift = int(sr * frame_shift_ms * 0.001)
    padded = 1 << (win - 1).bit_length()
    if x.size < win:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    m = 1 + (x.size - win) // shift
    idx = np.arange(win)[None, :] + shift * np.arange(m)[:, None]
    fr = x[idx].astype(np.float32)
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=np.float32)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
    fr = fr - np.float32(preemph) * prev
    fr = fr * povey_window(win)[None, :]
    fr = np.pad(fr, ((0, 0), (0, padded - win)))
    spec = np.fft.rfft(fr.astype(np.float32), axis=-1)
    power = (spec.real.astype(np.float32) ** 2 + spec.imag.astype(np.float32) ** 2)
    mel = power @ kaldi_mel_banks(num_mel_bins, padded, sr).T
    return np.log(np.maximum(mel, np.finfo(np.float32).eps)).astype(np.float32)
