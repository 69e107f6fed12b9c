"""`compute-fbank-feats` written FROM THE KALDI SPECIFICATION — test infrastructure, a second pin for SURVEY row a-4.

torchaudio (what touchnet/data/functions.py:117-134 calls) is not in this image and the reference holds no feature vector,
so `oracle/frontend.py::kaldi_fbank` (a restatement of torchaudio.compliance.kaldi.fbank) cannot be checked against its real
executor.  It is held to two implementations written independently of it and of each other:
  1. transformers.audio_utils' Kaldi-compatible fbank (third party; tests/golden/kaldi_fbank_hf.npz), and
  2. THIS file: the algorithm of Kaldi's own feature pipeline, followed step by step in float64 scalar loops with Kaldi's
     option structs and their defaults (the reference's overrides are passed by the caller, like on a Kaldi command line):
        FrameExtractionOptions  feat/feature-window.h     samp_freq 16000, frame_shift_ms 10, frame_length_ms 25,
                                                          dither 1.0, preemph_coeff 0.97, remove_dc_offset true,
                                                          window_type "povey", round_to_power_of_two true,
                                                          blackman_coeff 0.42, snip_edges true
        MelBanksOptions         feat/mel-computations.h   num_bins 25 (struct default; the fbank binary uses 23), low_freq 20,
                                                          high_freq 0 (= Nyquist), vtln off
        FbankOptions            feat/feature-fbank.h      use_energy false, energy_floor 0.0, raw_energy true,
                                                          htk_compat false, use_log_fbank true, use_power true
     Steps: NumFrames / ExtractWindow (snip_edges) -> ProcessWindow (dither, DC removal, pre-emphasis from the last sample
     down, window) -> zero-pad to the power of two -> power spectrum (ComputePowerSpectrum: bin 0 holds the DC energy, the
     Nyquist energy is dropped with the real-FFT packing) -> MelBanks (triangles in mel = 1127 ln(1 + f/700) over the FFT bins
     whose mel frequency lies strictly inside a triangle, see mel_banks) -> floor at FLT_EPSILON -> natural log.
Nothing here is imported by the product; tests compare the oracle and the HIP kernel with its output."""
import math

import numpy as np

FLT_EPSILON = 1.1920928955078125e-07


class FrameExtractionOptions:
    def __init__(self, **kw):
        self.samp_freq, self.frame_shift_ms, self.frame_length_ms = 16000.0, 10.0, 25.0
        self.dither, self.preemph_coeff, self.remove_dc_offset = 1.0, 0.97, True
        self.window_type, self.round_to_power_of_two, self.blackman_coeff, self.snip_edges = "povey", True, 0.42, True
        self.__dict__.update(kw)

    def window_shift(self):
        return int(self.samp_freq * 0.001 * self.frame_shift_ms)

    def window_size(self):
        return int(self.samp_freq * 0.001 * self.frame_length_ms)

    def padded_window_size(self):
        n = self.window_size()
        if not self.round_to_power_of_two:
            return n
        p = 1
        while p < n:
            p *= 2
        return p


class MelBanksOptions:
    def __init__(self, **kw):
        self.num_bins, self.low_freq, self.high_freq = 25, 20.0, 0.0
        self.__dict__.update(kw)


class FbankOptions:
    def __init__(self, frame_opts=None, mel_opts=None, **kw):
        self.frame_opts, self.mel_opts = frame_opts or FrameExtractionOptions(), mel_opts or MelBanksOptions()
        self.use_energy, self.energy_floor, self.raw_energy = False, 0.0, True
        self.htk_compat, self.use_log_fbank, self.use_power = False, True, True
        self.__dict__.update(kw)


def mel_scale(freq):
    return 1127.0 * math.log(1.0 + freq / 700.0)


def feature_window(opts):
    n = opts.window_size()
    a = 2.0 * math.pi / (n - 1)
    w = []
    for i in range(n):
        if opts.window_type == "hanning":
            w.append(0.5 - 0.5 * math.cos(a * i))
        elif opts.window_type == "hamming":
            w.append(0.54 - 0.46 * math.cos(a * i))
        elif opts.window_type == "povey":
            w.append(math.pow(0.5 - 0.5 * math.cos(a * i), 0.85))
        elif opts.window_type == "rectangular":
            w.append(1.0)
        elif opts.window_type == "blackman":
            w.append(opts.blackman_coeff - 0.5 * math.cos(a * i) + (0.5 - opts.blackman_coeff) * math.cos(2 * a * i))
        else:
            raise ValueError(opts.window_type)
    return w


def num_frames(num_samples, opts):
    shift, size = opts.window_shift(), opts.window_size()
    if opts.snip_edges:
        return 0 if num_samples < size else 1 + (num_samples - size) // shift
    return (num_samples + shift // 2) // shift


def mel_banks(opts, frame_opts):
    """MelBanks::MelBanks: list of (first FFT bin, weights) per mel bin.  A FFT bin contributes to a mel bin when its mel
    frequency lies STRICTLY inside (left, right); the weight rises on (left, center] and falls on (center, right)."""
    padded = frame_opts.padded_window_size()
    num_fft_bins = padded // 2
    nyquist = 0.5 * frame_opts.samp_freq
    high = opts.high_freq if opts.high_freq > 0.0 else nyquist + opts.high_freq
    if not (0.0 <= opts.low_freq < nyquist and 0.0 < high <= nyquist and opts.low_freq < high):
        raise ValueError("bad low/high frequency")
    fft_bin_width = frame_opts.samp_freq / padded
    mel_low, mel_high = mel_scale(opts.low_freq), mel_scale(high)
    delta = (mel_high - mel_low) / (opts.num_bins + 1)
    bins = []
    for b in range(opts.num_bins):
        left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
        first, weights = -1, []
        for i in range(num_fft_bins):
            mel = mel_scale(fft_bin_width * i)
            if left < mel < right:
                weights.append((mel - left) / (center - left) if mel <= center else (right - mel) / (right - center))
                if first < 0:
                    first = i
        if first < 0:
            raise ValueError("empty mel bin: too many bins for this window")
        bins.append((first, weights))
    return bins


def compute_fbank_feats(wave, opts):
    """wave: samples at the scale Kaldi reads from a 16-bit wav file (integers in [-32768, 32767]) -> [frames][num_bins]."""
    fo = opts.frame_opts
    if fo.dither != 0.0:
        raise ValueError("dither draws random numbers; every reference recipe sets --dither=0")
    if opts.use_energy or opts.htk_compat or not opts.use_power or not opts.use_log_fbank:
        raise ValueError("only the option set of the reference's recipes is written out here")
    wave = [float(v) for v in wave]
    size, shift, padded = fo.window_size(), fo.window_shift(), fo.padded_window_size()
    window, banks = feature_window(fo), mel_banks(opts.mel_opts, fo)
    out = []
    for f in range(num_frames(len(wave), fo)):
        x = wave[f * shift:f * shift + size]                               # ExtractWindow, snip_edges
        if fo.remove_dc_offset:                                            # ProcessWindow
            mean = sum(x) / size
            x = [v - mean for v in x]
        if fo.preemph_coeff != 0.0:
            for i in range(size - 1, 0, -1):
                x[i] -= fo.preemph_coeff * x[i - 1]
            x[0] -= fo.preemph_coeff * x[0]
        x = [v * w for v, w in zip(x, window)] + [0.0] * (padded - size)
        spec = np.fft.rfft(np.asarray(x, dtype=np.float64))                # srfft + ComputePowerSpectrum
        power = (spec.real ** 2 + spec.imag ** 2)[:padded // 2]            # (the Nyquist bin is not part of the mel input)
        row = []
        for first, weights in banks:                                       # MelBanks::Compute
            e = 0.0
            for j, w in enumerate(weights):
                e += w * power[first + j]
            row.append(math.log(max(e, FLT_EPSILON)))
        out.append(row)
    return np.asarray(out, dtype=np.float64).reshape(len(out), opts.mel_opts.num_bins)


def reference_recipe_fbank(pcm_int16, num_mel_bins=80, sample_rate=16000.0):
    """The option overrides of touchnet/data/functions.py:117-134 on Kaldi's defaults: num_mel_bins, frame length / shift,
    dither 0, energy_floor 0, sample_frequency; input at int16 scale (the reference multiplies by 1 << 15)."""
    opts = FbankOptions(FrameExtractionOptions(samp_freq=float(sample_rate), dither=0.0, frame_length_ms=25.0,
                                               frame_shift_ms=10.0),
                        MelBanksOptions(num_bins=num_mel_bins), energy_floor=0.0)
    return compute_fbank_feats(pcm_int16, opts)
