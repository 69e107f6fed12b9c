// Audio frontend on the GPU: waveform -> Kaldi log-fbank / Whisper log-mel -> stack+stride+normalise.
//
// Replaces the CPU dataloader-worker stages
//   audio_compute_fbank               touchnet/data/functions.py:117-134 (torchaudio.compliance.kaldi.fbank,
//                                     energy_floor=0.0, dither=0.0, waveform * 32768)
//   audio_compute_log_mel_spectrogram touchnet/data/functions.py:159-190 (torch.stft hann-400 hop-160
//                                     center/reflect, drop last frame, slaney mel, log10, max-8, (x+4)/4);
//                                     same numbers as HF WhisperFeatureExtractor used by
//                                     touchnet/models/qwen2_audio/processing_qwen2_audio.py:63-70
//   audiofeat_stack                   touchnet/data/functions.py:258-286
// so the dataloader can hand raw PCM to the device (SURVEY.md §8b hook 4).  16 kHz only (every recipe).
//
// All three are HBM-trivial (320 B in, 320-512 B out per 10 ms frame); the kernels keep one frame's
// working set in LDS and are bounded by LDS/VALU, not memory: fbank = 512-point radix-2 FFT in LDS,
// log-mel = 400-point direct DFT (400 is not a power of two; 4 frames share each twiddle load).
#include "common.h"
#include <float.h>

namespace tn {

constexpr int kWin = 400, kShift = 160, kPad = 512, kBins = 257;
constexpr float kSr = 16000.f;

__device__ __forceinline__ int bitrev9(int x) { return (int)(__brev((unsigned)x) >> 23); }

__global__ __launch_bounds__(256) void kaldi_fbank_kernel(const float* __restrict__ wav, float* __restrict__ feat,
                                                          int n_frames, int n_mels) {
  __shared__ float re[kPad], im[kPad], twc[kPad / 2], tws[kPad / 2], win[kWin], melk[kBins + 3], pw[kBins + 3];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  {  // per-block constants
    float s, c;
    sincospif((float)tid / 256.f, &s, &c);  // angle = 2*pi*tid/512
    twc[tid] = c;
    tws[tid] = -s;
    for (int n = tid; n < kWin; n += 256)
      win[n] = powf(0.5f - 0.5f * cospif(2.f * (float)n / (float)(kWin - 1)), 0.85f);
    for (int k = tid; k < kBins; k += 256) melk[k] = 1127.f * logf(1.f + (kSr / kPad) * (float)k / 700.f);
  }
  const float mel_lo = 1127.f * logf(1.f + 20.f / 700.f);
  const float mel_hi = 1127.f * logf(1.f + (0.5f * kSr) / 700.f);
  const float mdelta = (mel_hi - mel_lo) / (float)(n_mels + 1);
  __syncthreads();

  for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
    const float* src = wav + (size_t)f * kShift;
    float part = 0.f;
    for (int n = tid; n < kWin; n += 256) {
      const float x = src[n] * 32768.f;
      im[n] = x;  // raw samples parked in `im`
      part += x;
    }
    const float mean = block_sum(part, red) / (float)kWin;  // (contains the barriers that publish im[])
    for (int n = tid; n < kPad; n += 256) {
      float y = 0.f;
      if (n < kWin) {
        const float cur = im[n] - mean;
        const float prev = (n > 0 ? im[n - 1] : im[0]) - mean;
        y = (cur - 0.97f * prev) * win[n];
      }
      re[bitrev9(n)] = y;
    }
    __syncthreads();
    for (int n = tid; n < kPad; n += 256) im[n] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int s = 1; s <= 9; ++s) {
      const int half = 1 << (s - 1);
      const int pos = tid & (half - 1);
      const int i = ((tid >> (s - 1)) << s) + pos;
      const int j = i + half;
      const int tw = pos << (9 - s);
      const float c = twc[tw], sn = tws[tw];
      const float br = re[j] * c - im[j] * sn, bi = re[j] * sn + im[j] * c;
      const float ar = re[i], ai = im[i];
      re[i] = ar + br;
      im[i] = ai + bi;
      re[j] = ar - br;
      im[j] = ai - bi;
      __syncthreads();
    }
    for (int k = tid; k < kBins; k += 256) pw[k] = re[k] * re[k] + im[k] * im[k];
    __syncthreads();
    if (tid < n_mels) {
      const float left = mel_lo + (float)tid * mdelta, center = left + mdelta, right = center + mdelta;
      float acc = 0.f;
      for (int k = 0; k < kPad / 2; ++k) {  // bin 256 carries zero weight (torchaudio pads one zero column)
        const float m = melk[k];
        const float w = fminf((m - left) / (center - left), (right - m) / (right - center));
        if (w > 0.f) acc += w * pw[k];
      }
      feat[(size_t)f * n_mels + tid] = logf(fmaxf(acc, FLT_EPSILON));
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------- log-mel
constexpr int kNfft = 400, kHop = 160, kFreq = 201, kFr = 4;

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void fill_kernel(float* p, float v) { p[0] = v; }

__global__ __launch_bounds__(256) void log_mel_kernel(const float* __restrict__ wav, const float* __restrict__ fb,
                                                      float* __restrict__ feat, float* __restrict__ gmax,
                                                      int n_samples, int n_frames, int n_mels) {
  __shared__ float twc[kNfft], tws[kNfft], hann[kNfft], xs[kFr][kNfft], pws[kFr][kFreq + 3];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  for (int n = tid; n < kNfft; n += 256) {
    float s, c;
    sincospif(2.f * (float)n / (float)kNfft, &s, &c);
    twc[n] = c;
    tws[n] = s;
    hann[n] = 0.5f - 0.5f * c;  // periodic hann: 0.5 - 0.5 cos(2 pi n / 400)
  }
  __syncthreads();
  float lmax = -INFINITY;
  const int n_groups = (n_frames + kFr - 1) / kFr;
  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int t0 = grp * kFr;
    for (int e = tid; e < kFr * kNfft; e += 256) {
      const int f = e / kNfft, n = e % kNfft;
      int src = (t0 + f) * kHop + n - kNfft / 2;  // center=True, reflect padding
      if (src < 0) src = -src;
      if (src >= n_samples) src = 2 * (n_samples - 1) - src;
      const float x = (t0 + f < n_frames && src >= 0 && src < n_samples) ? wav[src] : 0.f;
      xs[f][n] = x * hann[n];
    }
    __syncthreads();
    if (tid < kFreq) {
      float re[kFr], im[kFr];
#pragma unroll
      for (int f = 0; f < kFr; ++f) re[f] = im[f] = 0.f;
      int idx = 0;
      for (int n = 0; n < kNfft; ++n) {
        const float c = twc[idx], s = tws[idx];
#pragma unroll
        for (int f = 0; f < kFr; ++f) {
          const float x = xs[f][n];
          re[f] += x * c;
          im[f] -= x * s;
        }
        idx += tid;
        if (idx >= kNfft) idx -= kNfft;
      }
#pragma unroll
      for (int f = 0; f < kFr; ++f) pws[f][tid] = re[f] * re[f] + im[f] * im[f];
    }
    __syncthreads();
    for (int e = tid; e < kFr * n_mels; e += 256) {
      const int f = e / n_mels, m = e % n_mels;
      if (t0 + f < n_frames) {
        const float* w = fb + (size_t)m * kFreq;
        float acc = 0.f;
        for (int k = 0; k < kFreq; ++k) acc += w[k] * pws[f][k];
        const float v = log10f(fmaxf(acc, 1e-10f));
        feat[(size_t)(t0 + f) * n_mels + m] = v;
        lmax = fmaxf(lmax, v);
      }
    }
    __syncthreads();
  }
  lmax = block_max(lmax, red);
  if (tid == 0 && lmax > -INFINITY) atomic_max_float(gmax, lmax);
}

__global__ void log_mel_finish_kernel(float* __restrict__ feat, const float* __restrict__ gmax, size_t n) {
  const float floor_v = gmax[0] - 8.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    feat[i] = (fmaxf(feat[i], floor_v) + 4.f) * 0.25f;
}

// ---------------------------------------------------------------------------------------------- stack
__global__ __launch_bounds__(256) void audiofeat_stack_kernel(const float* __restrict__ feat, float* __restrict__ out,
                                                              int T, int F, int stack, int stride, int t_lfr,
                                                              int normalize) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= t_lfr) return;
  const int n = stack * F, lp = (stack - 1) / 2;
  auto src = [&](int e) {
    const int j = e / F, f = e % F;
    int t = row * stride + j - lp;  // left pad = copies of frame 0, right pad = copies of the last frame
    t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
    return feat[(size_t)t * F + f];
  };
  float mean = 0.f, inv = 1.f;
  if (normalize) {
    float s = 0.f;
    for (int e = lane; e < n; e += 64) s += src(e);
    mean = wave_sum(s) / (float)n;
    float ss = 0.f;
    for (int e = lane; e < n; e += 64) {
      const float d = src(e) - mean;
      ss += d * d;
    }
    const float var = wave_sum(ss) / (float)(n - 1);  // unbiased, as torch.std
    inv = 1.f / (sqrtf(var) + 1e-5f);
  }
  for (int e = lane; e < n; e += 64) out[(size_t)row * n + e] = (src(e) - mean) * inv;
}

// ---------------------------------------------------------------------------------------------- feature-level augmentation
// touchnet/data/functions.py:193-255 (audiofeat_spec_aug, audiofeat_spec_sub, audiofeat_spec_trim) as ONE gather pass over
// the [T, F] feature matrix: the random draws are the host's (same `random` calls in the same order as the reference's
// stage functions, touchnet_amd/data/functions.py), the device applies them:
//   y[t][f] = 0                      if row src(t) lies in a time stripe or f in a frequency stripe   (spec_aug :205-217)
//           = x[src(t)][f]           with src(t) = t - pos_k for the LAST substitution k whose rows [start_k, end_k)
//                                    contain t (later substitutions overwrite earlier ones, all read the stage's
//                                    INPUT, :233-239), else t
//   t < out_rows                     (spec_trim drops the tail rows, :250-253)
// spec_aug runs in front of spec_sub in the reference's chain (processing_touch_audio.py:468-473), so the stripes are
// tested on the SOURCE row.  HBM-bound: 8 B per element, one launch per utterance.
struct AugPlan {
  int n_t, n_f, n_s;
  int t[16][2], f[16][2], s[16][3];
};

__global__ __launch_bounds__(256) void feat_augment_kernel(const float* __restrict__ x, float* __restrict__ y, int F,
                                                           long long n, AugPlan plan) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i / F), f = (int)(i % F);
  int src = t;
  for (int k = plan.n_s - 1; k >= 0; --k)
    if (t >= plan.s[k][0] && t < plan.s[k][1]) {
      src = t - plan.s[k][2];
      break;
    }
  bool zero = false;
  for (int k = 0; k < plan.n_t; ++k) zero |= (src >= plan.t[k][0] && src < plan.t[k][1]);
  for (int k = 0; k < plan.n_f; ++k) zero |= (f >= plan.f[k][0] && f < plan.f[k][1]);
  y[i] = zero ? 0.f : x[(size_t)src * F + f];
}

// ---------------------------------------------------------------------------------------------- speed perturbation
// touchnet/data/functions.py:99-114: sox `speed s` + `rate sr` = the waveform resampled by the factor 1/s (pitch and tempo
// change together).  libsox's polyphase resampler is a third-party binary algorithm that cannot be restated bit for bit
// (and torchaudio is not in this image to compare with): this kernel evaluates the band-limited interpolation
//     y[n] = sum_k x[k] h(n s - k),   h(t) = c sinc(c t) kaiser(t / W),  c = 0.95 min(1, 1/s)
// from a polyphase table the host builds in float64 for the rational speed p / q (touchnet_amd/functional.py): output n sits
// at input position (n p) / q -> integer part i, phase r = (n p) mod q, and y[n] = sum_j x[i - half + 1 + j] tab[r][j].
// PARITY UNPINNED against libsox (stated in DESIGN.md); held to a float64 evaluation of the same formula (oracle/frontend.py).
__global__ __launch_bounds__(256) void resample_polyphase_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 const float* __restrict__ tab, long long n_in,
                                                                 long long n_out, int p, int q, int ntap) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  const long long pos = n * p;
  const long long i = pos / q;
  const int r = (int)(pos - i * q);
  const float* t = tab + (size_t)r * ntap;
  const long long k0 = i - ntap / 2 + 1;
  float acc = 0.f;
  for (int j = 0; j < ntap; ++j) {
    const long long k = k0 + j;
    const float xv = (k >= 0 && k < n_in) ? x[k] : 0.f;
    acc = fmaf(xv, t[j], acc);
  }
  y[n] = acc;
}

// int16 PCM -> float32 in [-1, 1): x * 2^-15, exactly numpy's `astype(float32) / 32768.0` of the reference's
// datapipe (touchnet/data/datapipe.py:164).  Lets the caller upload 2 bytes per sample.
__global__ __launch_bounds__(256) void pcm16_to_f32_kernel(const int16_t* __restrict__ in, float* __restrict__ out,
                                                           long long n) {
  const long long i8 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i8 >= n) return;
  if (i8 + 8 <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + i8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = (float)(int16_t)(w[k] & 0xffffu) * (1.f / 32768.f);
      f[2 * k + 1] = (float)(int16_t)(w[k] >> 16) * (1.f / 32768.f);
    }
    float4* o = reinterpret_cast<float4*>(out + i8);
    o[0] = make_float4(f[0], f[1], f[2], f[3]);
    o[1] = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    for (long long i = i8; i < min(i8 + 8, n); ++i) out[i] = (float)in[i] * (1.f / 32768.f);
  }
}

}  // namespace tn

using namespace tn;

extern "C" {

int tn_fbank_frames(int n_samples) { return n_samples < kWin ? 0 : 1 + (n_samples - kWin) / kShift; }

int tn_kaldi_fbank(const float* wav, float* feat, int n_samples, int n_mels, void* stream) {
  const int nf = tn_fbank_frames(n_samples);
  if (n_mels <= 0 || n_mels > 256) return TN_EINVAL;
  if (nf == 0) return TN_OK;
  hipLaunchKernelGGL(kaldi_fbank_kernel, dim3(nf < 2048 ? nf : 2048), dim3(256), 0, (hipStream_t)stream, wav, feat,
                     nf, n_mels);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// scratch_max: device float[1]
int tn_log_mel(const float* wav, const float* mel_fb, float* feat, float* scratch_max, int n_samples, int n_mels,
               void* stream) {
  if (n_mels <= 0 || n_samples <= kNfft / 2) return TN_EINVAL;
  const int nf = n_samples / kHop;
  if (nf == 0) return TN_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(1), 0, st, scratch_max, -INFINITY);
  const int groups = (nf + kFr - 1) / kFr;
  hipLaunchKernelGGL(log_mel_kernel, dim3(groups < 1024 ? groups : 1024), dim3(256), 0, st, wav, mel_fb, feat,
                     scratch_max, n_samples, nf, n_mels);
  const size_t n = (size_t)nf * n_mels;
  hipLaunchKernelGGL(log_mel_finish_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256),
                     0, st, feat, scratch_max, n);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_audiofeat_stack(const float* feat, float* out, int T, int F, int stack, int stride, int normalize,
                       void* stream) {
  if (T <= 0 || F <= 0 || stack <= 0 || stride <= 0) return TN_EINVAL;
  const int t_lfr = (T + stride - 1) / stride;
  hipLaunchKernelGGL(audiofeat_stack_kernel, dim3((t_lfr + 3) / 4), dim3(256), 0, (hipStream_t)stream, feat, out, T,
                     F, stack, stride, t_lfr, normalize);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// x [T, F] -> y [out_rows, F] (out_rows <= T).  HOST arrays: t_masks / f_masks = n x [start, end), subs = n x (start, end,
// pos) with 0 <= pos <= start; at most 16 of each.
int tn_feat_augment(const float* x, float* y, int T, int F, int out_rows, const int* t_masks, int n_t,
                    const int* f_masks, int n_f, const int* subs, int n_sub, void* stream) {
  if (T <= 0 || F <= 0 || out_rows <= 0 || out_rows > T || n_t < 0 || n_f < 0 || n_sub < 0 || n_t > 16 || n_f > 16 ||
      n_sub > 16 || x == y)
    return TN_EINVAL;
  AugPlan plan;
  plan.n_t = n_t, plan.n_f = n_f, plan.n_s = n_sub;
  for (int k = 0; k < n_t; ++k) plan.t[k][0] = t_masks[2 * k], plan.t[k][1] = t_masks[2 * k + 1];
  for (int k = 0; k < n_f; ++k) plan.f[k][0] = f_masks[2 * k], plan.f[k][1] = f_masks[2 * k + 1];
  for (int k = 0; k < n_sub; ++k) {
    plan.s[k][0] = subs[3 * k], plan.s[k][1] = subs[3 * k + 1], plan.s[k][2] = subs[3 * k + 2];
    if (plan.s[k][2] < 0 || plan.s[k][2] > plan.s[k][0] || plan.s[k][1] > T) return TN_EINVAL;   // source row stays inside x
  }
  const long long n = (long long)out_rows * F;
  hipLaunchKernelGGL(feat_augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, F,
                     n, plan);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// x [n_in] -> y [n_out] at the rational rate p / q input samples per output sample; tab: device float [q][ntap] (ntap even).
int tn_resample_polyphase(const float* x, float* y, const float* tab, long long n_in, long long n_out, int p, int q,
                          int ntap, void* stream) {
  if (n_in <= 0 || n_out <= 0 || p <= 0 || q <= 0 || ntap <= 0 || (ntap & 1) || x == y) return TN_EINVAL;
  hipLaunchKernelGGL(resample_polyphase_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, y, tab, n_in, n_out, p, q, ntap);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_pcm16_to_f32(const void* pcm, float* out, long long n, void* stream) {
  if (n <= 0) return TN_OK;
  const long long blocks = (n + 2047) / 2048;
  if (blocks > 0x7fffffffLL) return TN_EINVAL;
  hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const int16_t*)pcm, out, n);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

}  // extern "C"
