// Fused gradient-norm + clip + AdamW over FLAT fp32 master / bf16 shadow buffers.
//
// Replaces, on the single-GPU path and on every FSDP shard,
//   touchnet/utils/distributed.py:426-491   clip_grad_norm_ (foreach L2 norm, clip coefficient)
//   touchnet/bin/train.py:458-474           NaN/Inf grad-norm => skip the step (a host sync there)
//   touchnet/utils/optimizer.py:157-172     torch.optim.AdamW(betas=(0.9,0.95), wd=0.1, fused=True)
// with two passes over HBM: (1) sum of squares, deterministic two-stage; (2) one read of g/p/m/v and one
// write of p/m/v (+ bf16 shadow of p for the next forward).  28 B/param algorithmic traffic with a bf16
// gradient: g 2 + p,m,v 12 read, p,m,v 12 + shadow 2 written.
#include "common.h"

namespace tn {

constexpr int kSumsqBlocks = 1024;

template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* __restrict__ g, float* __restrict__ partial,
                                                            size_t n) {
  constexpr int N = Vec16<T>::N;
  __shared__ float sm[4];
  float acc = 0.f;
  const size_t nvec = n / N;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    Vec16<T> a;
    float f[N];
    a.load(g + v * N);
    a.unpack(f);
#pragma unroll
    for (int j = 0; j < N; ++j) acc += f[j] * f[j];
  }
  if (blockIdx.x == 0)
    for (size_t i = nvec * N + threadIdx.x; i < n; i += blockDim.x) {
      const float f = Elem<T>::ld(g + i);
      acc += f * f;
    }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ norm_sq, int nb) {
  __shared__ float sm[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[i];
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) norm_sq[0] += acc;
}

// Multi-tensor form: ONE launch over all gradient tensors of a dtype.  Workgroup b owns chunk b: it finds its tensor
// by binary search in the (ascending) first-chunk table and reduces kSumsqChunk elements of it.  The per-tensor
// form above costs two launches per tensor: 1750 launches of 3-7 us for the 875 tensors of a 7B model.
constexpr long long kSumsqChunk = 32768;

template <typename T>
__global__ __launch_bounds__(256) void sumsq_multi_kernel(const void* const* __restrict__ ptrs,
                                                          const long long* __restrict__ sizes,
                                                          const long long* __restrict__ first_chunk, int ntensors,
                                                          float* __restrict__ partial) {
  constexpr int N = Vec16<T>::N;
  __shared__ float sm[4];
  const long long c = blockIdx.x;
  int lo = 0, hi = ntensors - 1;                     // last t with first_chunk[t] <= c
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_chunk[mid] <= c) lo = mid; else hi = mid - 1;
  }
  const T* g = static_cast<const T*>(ptrs[lo]);
  const long long beg = (c - first_chunk[lo]) * kSumsqChunk;
  const long long end = min(beg + kSumsqChunk, sizes[lo]);
  float acc = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {  // (chunk starts are multiples of 32768 elements: aligned with g)
    const long long nvec = (end - beg) / N;
    for (long long v = threadIdx.x; v < nvec; v += 256) {
      Vec16<T> a;
      float f[N];
      a.load(g + beg + v * N);
      a.unpack(f);
#pragma unroll
      for (int j = 0; j < N; ++j) acc += f[j] * f[j];
    }
    for (long long i = beg + nvec * N + threadIdx.x; i < end; i += 256) {
      const float f = Elem<T>::ld(g + i);
      acc += f * f;
    }
  } else {
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
      const float f = Elem<T>::ld(g + i);
      acc += f * f;
    }
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[c] = acc;
}

template <typename G> __device__ __forceinline__ void load4(const G* g, size_t i, float (&f)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* g, size_t i, float (&f)[4]) {
  const float4 t = reinterpret_cast<const float4*>(g)[i];
  f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* g, size_t i, float (&f)[4]) {
  const uint2 t = reinterpret_cast<const uint2*>(g)[i];
  f[0] = __uint_as_float(t.x << 16); f[1] = __uint_as_float(t.x & 0xffff0000u);
  f[2] = __uint_as_float(t.y << 16); f[3] = __uint_as_float(t.y & 0xffff0000u);
}

template <typename G>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m,
                                                    float* __restrict__ v, const G* __restrict__ g,
                                                    bf16_t* __restrict__ shadow, const float* __restrict__ norm_sq,
                                                    size_t n, float lr, float b1, float b2, float eps, float wd,
                                                    float max_norm, float bc1, float bc2) {
  const float nsq = norm_sq ? norm_sq[0] : 0.f;
  if (!(nsq == nsq) || nsq > 3.0e38f) return;  // NaN / Inf total norm: skip the whole step
  float clip = 1.f;
  if (norm_sq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(nsq) + 1e-6f));
  const float step = lr / bc1, rs_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float gf[4];
    load4<G>(g, i, gf);
#pragma unroll
    for (int j = 0; j < 4; ++j) gf[j] *= clip;
    float* pp = &pv.x;
    float* mp = &mv.x;
    float* vp = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pp[j] *= decay;
      mp[j] = b1 * mp[j] + (1.f - b1) * gf[j];
      vp[j] = b2 * vp[j] + (1.f - b2) * gf[j] * gf[j];
      pp[j] -= step * mp[j] / (sqrtf(vp[j]) * rs_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (shadow) {
      uint2 s;
      s.x = pack2bf(pp[0], pp[1]);
      s.y = pack2bf(pp[2], pp[3]);
      reinterpret_cast<uint2*>(shadow)[i] = s;
    }
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
      const float gi = Elem<G>::ld(g + i) * clip;
      float pi = p[i] * decay;
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      pi -= step * mi / (sqrtf(vi) * rs_bc2 + eps);
      p[i] = pi;
      m[i] = mi;
      v[i] = vi;
      if (shadow) shadow[i] = f2bf(pi);
    }
}

// ---- multi-tensor AdamW: ONE launch for all parameters whose gradient has the same dtype -----------------------
// (the per-tensor entry point above costs a launch per tensor: 875 launches of a few us for a 7B model, and on FSDP
// shards — 1/8 of each tensor — the launches would dominate the pass).  Workgroup c owns chunk c of kAdamChunk
// elements; it finds its tensor by binary search in the ascending first-chunk table, like sumsq_multi_kernel.
// Step count, bias corrections, clip coefficient and the skip decision live in a small DEVICE state written by
// adamw_prepare_kernel: the step only advances when the gradient norm is finite (torch's AdamW does not advance on a
// skipped step either), with no host round trip.
constexpr long long kAdamChunk = 16384;

// state: [0] step (int bits) [1] bias_corr1 [2] bias_corr2 [3] clip coefficient [4] skip (1.0 = non-finite norm)
__global__ void adamw_prepare_kernel(const float* __restrict__ norm_sq, float* __restrict__ state, float b1, float b2,
                                     float max_norm) {
  const float nsq = norm_sq ? norm_sq[0] : 0.f;
  const bool bad = !(nsq == nsq) || nsq > 3.0e38f;
  int step = __float_as_int(state[0]);
  if (!bad) ++step;
  state[0] = __int_as_float(step);
  const float fs = (float)(step < 1 ? 1 : step);
  state[1] = 1.f - powf(b1, fs);
  state[2] = 1.f - powf(b2, fs);
  state[3] = (norm_sq && max_norm > 0.f && !bad) ? fminf(1.f, max_norm / (sqrtf(nsq) + 1e-6f)) : 1.f;
  state[4] = bad ? 1.f : 0.f;
}

template <typename G>
__global__ __launch_bounds__(256) void adamw_multi_kernel(float* const* __restrict__ ps, float* const* __restrict__ ms,
                                                          float* const* __restrict__ vs,
                                                          const void* const* __restrict__ gs,
                                                          bf16_t* const* __restrict__ shadows,
                                                          const long long* __restrict__ sizes,
                                                          const long long* __restrict__ first_chunk, int ntensors,
                                                          long long nchunks,
                                                          const float* __restrict__ state, float lr, float b1, float b2,
                                                          float eps, float wd) {
  if (state[4] != 0.f) return;                       // NaN / Inf total norm: the whole step is skipped
  // one chunk per workgroup — or, with a bounded grid (tn_adamw_multi_bounded), a stride of gridDim.x chunks: the few
  // resident workgroups then take a bounded share of the HBM bandwidth beside another stream's kernels
  for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
  int lo = 0, hi = ntensors - 1;                     // last t with first_chunk[t] <= c
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first_chunk[mid] <= c) lo = mid; else hi = mid - 1;
  }
  float* p = ps[lo];
  float* m = ms[lo];
  float* v = vs[lo];
  const G* g = static_cast<const G*>(gs[lo]);
  bf16_t* shadow = shadows[lo];
  const long long beg = (c - first_chunk[lo]) * kAdamChunk;
  const long long end = min(beg + kAdamChunk, sizes[lo]);
  const float clip = state[3];
  const float step = lr / state[1], rs_bc2 = rsqrtf(state[2]), decay = 1.f - lr * wd;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(g) |
                         reinterpret_cast<uintptr_t>(shadow)) & 15) == 0;
  long long done = beg;
  if (aligned) {
    const long long n4 = (end - beg) / 4;
    const size_t base4 = (size_t)(beg / 4);          // beg is a multiple of kAdamChunk
    for (long long i = threadIdx.x; i < n4; i += 256) {
      float4 pv = reinterpret_cast<float4*>(p)[base4 + i];
      float4 mv = reinterpret_cast<float4*>(m)[base4 + i];
      float4 vv = reinterpret_cast<float4*>(v)[base4 + i];
      float gf[4];
      load4<G>(g, base4 + i, gf);
      float* pp = &pv.x;
      float* mp = &mv.x;
      float* vp = &vv.x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = gf[j] * clip;
        pp[j] *= decay;
        mp[j] = b1 * mp[j] + (1.f - b1) * gj;
        vp[j] = b2 * vp[j] + (1.f - b2) * gj * gj;
        pp[j] -= step * mp[j] / (sqrtf(vp[j]) * rs_bc2 + eps);
      }
      reinterpret_cast<float4*>(p)[base4 + i] = pv;
      reinterpret_cast<float4*>(m)[base4 + i] = mv;
      reinterpret_cast<float4*>(v)[base4 + i] = vv;
      if (shadow) reinterpret_cast<uint2*>(shadow)[base4 + i] = make_uint2(pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]));
    }
    done = beg + n4 * 4;
  }
  for (long long i = done + threadIdx.x; i < end; i += 256) {
    const float gi = Elem<G>::ld(g + i) * clip;
    float pi = p[i] * decay;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    pi -= step * mi / (sqrtf(vi) * rs_bc2 + eps);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (shadow) shadow[i] = f2bf(pi);
  }
  }
}

}  // namespace tn

using namespace tn;

extern "C" {

int tn_sumsq_scratch_floats(void) { return kSumsqBlocks; }

// norm_sq[0] += sum(g^2); scratch: kSumsqBlocks floats.  Stream-ordered, deterministic.
int tn_sumsq(const void* g, float* scratch, float* norm_sq, long long n, int dtype, void* stream) {
  if (n <= 0) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int nb = (int)((n / 8 + 255) / 256);
  nb = nb < 1 ? 1 : (nb > kSumsqBlocks ? kSumsqBlocks : nb);
  if (dtype == 0)
    hipLaunchKernelGGL((sumsq_partial_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)g, scratch,
                       (size_t)n);
  else if (dtype == 1)
    hipLaunchKernelGGL((sumsq_partial_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)g, scratch,
                       (size_t)n);
  else
    return TN_EINVAL;
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, st, scratch, norm_sq, nb);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

long long tn_sumsq_multi_chunk(void) { return kSumsqChunk; }

// norm_sq[0] += sum over `ntensors` tensors of sum(g^2).  Device tables: ptrs[t], sizes[t] (elements, > 0),
// first_chunk[t] = sum_{u<t} ceil(sizes[u] / chunk); `partial`: nchunks floats.  Deterministic (fixed chunking,
// fixed-order final reduction).
int tn_sumsq_multi(const void* const* ptrs, const long long* sizes, const long long* first_chunk, int ntensors,
                   long long nchunks, float* partial, float* norm_sq, int dtype, void* stream) {
  if (ntensors <= 0 || nchunks <= 0 || nchunks > 0x7fffffffLL) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL((sumsq_multi_kernel<float>), dim3((unsigned)nchunks), dim3(256), 0, st, ptrs, sizes, first_chunk,
                       ntensors, partial);
  else if (dtype == 1)
    hipLaunchKernelGGL((sumsq_multi_kernel<bf16_t>), dim3((unsigned)nchunks), dim3(256), 0, st, ptrs, sizes,
                       first_chunk, ntensors, partial);
  else
    return TN_EINVAL;
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, st, partial, norm_sq, (int)nchunks);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_adamw_step(float* p, float* m, float* v, const void* g, void* p_shadow_bf16, const float* norm_sq,
                  long long n, float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                  float bias_corr1, float bias_corr2, int g_dtype, void* stream) {
  if (n <= 0) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  size_t nb = ((size_t)n / 4 + 255) / 256;
  nb = nb < 1 ? 1 : (nb > 8192 ? 8192 : nb);
  if (g_dtype == 0)
    hipLaunchKernelGGL((adamw_kernel<float>), dim3((int)nb), dim3(256), 0, st, p, m, v, (const float*)g,
                       (bf16_t*)p_shadow_bf16, norm_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, max_norm,
                       bias_corr1, bias_corr2);
  else if (g_dtype == 1)
    hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3((int)nb), dim3(256), 0, st, p, m, v, (const bf16_t*)g,
                       (bf16_t*)p_shadow_bf16, norm_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, max_norm,
                       bias_corr1, bias_corr2);
  else
    return TN_EINVAL;
  TN_LAUNCH_CHECK();
  return TN_OK;
}

long long tn_adamw_multi_chunk(void) { return kAdamChunk; }

// state: 8 floats, zero-initialised once by the caller and owned by the optimizer (step count lives in it).
int tn_adamw_prepare(const float* norm_sq, float* state, float beta1, float beta2, float max_norm, void* stream) {
  if (state == nullptr) return TN_EINVAL;
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, norm_sq, state, beta1, beta2,
                     max_norm);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// Device tables of `ntensors` entries: p / m / v (fp32), g (g_dtype), shadow (bf16 copy of p or NULL), sizes
// (elements, > 0), first_chunk[t] = sum_{u<t} ceil(sizes[u] / tn_adamw_multi_chunk()).  Uses the state written by
// tn_adamw_prepare on the same stream.
int tn_adamw_multi_bounded(void* const* ps, void* const* ms, void* const* vs, const void* const* gs,
                           void* const* shadows, const long long* sizes, const long long* first_chunk, int ntensors,
                           long long nchunks, const float* state, float lr, float beta1, float beta2, float eps,
                           float weight_decay, int g_dtype, int max_workgroups, void* stream) {
  if (ntensors <= 0 || nchunks <= 0 || nchunks > 0x7fffffffLL || state == nullptr || max_workgroups < 0) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)((max_workgroups > 0 && max_workgroups < nchunks) ? max_workgroups : nchunks);
  if (g_dtype == 0)
    hipLaunchKernelGGL((adamw_multi_kernel<float>), dim3(grid), dim3(256), 0, st, (float* const*)ps,
                       (float* const*)ms, (float* const*)vs, gs, (bf16_t* const*)shadows, sizes, first_chunk, ntensors,
                       nchunks, state, lr, beta1, beta2, eps, weight_decay);
  else if (g_dtype == 1)
    hipLaunchKernelGGL((adamw_multi_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (float* const*)ps,
                       (float* const*)ms, (float* const*)vs, gs, (bf16_t* const*)shadows, sizes, first_chunk, ntensors,
                       nchunks, state, lr, beta1, beta2, eps, weight_decay);
  else
    return TN_EINVAL;
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_adamw_multi(void* const* ps, void* const* ms, void* const* vs, const void* const* gs, void* const* shadows,
                   const long long* sizes, const long long* first_chunk, int ntensors, long long nchunks,
                   const float* state, float lr, float beta1, float beta2, float eps, float weight_decay, int g_dtype,
                   void* stream) {
  return tn_adamw_multi_bounded(ps, ms, vs, gs, shadows, sizes, first_chunk, ntensors, nchunks, state, lr, beta1, beta2,
                                eps, weight_decay, g_dtype, 0, stream);
}

}  // extern "C"
