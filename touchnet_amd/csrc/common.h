// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of touchnet_amd.
// wave = 64 lanes; all kernels assume blockDim.x is a multiple of 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TN_OK 0
#define TN_EINVAL (-22)

#define TN_LAUNCH_CHECK()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return (int)e__;                 \
  } while (0)

namespace tn {

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// round-to-nearest-even (same as torch's float->bfloat16); lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

// raw v_exp_f32 (2^x, ~1 ulp, flushes denormal results): the softmax inner loops must not pay for the
// denormal fix-up sequence libm's exp2f adds
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 1 / (1 + e^-x) with v_exp_f32 + v_rcp_f32 (1 ulp each; no IEEE division sequence): ONE definition for the SwiGLU row
// kernels (norm_act.hip) and the fused SwiGLU epilogues of the GEMM (gemm.hip), which must agree bit for bit
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// Exact-erf GELU, 0.5 x (1 + erf(x / sqrt 2)) (transformers' ACT2FN["gelu"] of the Whisper / Qwen2-Audio encoder layers:
// touchnet/models/qwen2_audio/__init__.py drives WhisperEncoderLayer), and its derivative — ONE definition for the row
// kernels (norm_act.hip) and the GELU epilogues of the GEMM (gemm.hip EPI_GELU_FWD / _BWD), which must agree bit for bit.
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: three orders below bf16 resolution): 1 - erf(z) = t (a1 + t (a2 + t
// (a3 + t (a4 + t a5)))) e^{-z^2}, t = 1 / (1 + p z) — ~15 VALU instructions with two transcendentals instead of libm's
// erff (~40 with branches: too slow beside MFMAs, profiles/r05*), and e^{-z^2} = e^{-x^2 / 2} is the density's exponential too.
// ONE implementation, on PAIRS (v_pk_mul_f32 / v_pk_fma_f32: two IEEE operations per instruction — an epilogue's VALU time is
// exposed, profiles/r06h_*): the scalar forms below run it on (x, 0).  Every product / sum is written out and contraction is off
// inside, so that every caller — row kernel or GEMM epilogue, whatever is inlined around it — gets the same bits.
typedef __attribute__((ext_vector_type(2))) float gelu_f32x2;
__device__ __forceinline__ void gelu_parts2(gelu_f32x2 x, gelu_f32x2& cdf, gelu_f32x2& e) {
#pragma clang fp contract(off)
  const gelu_f32x2 ax = {fabsf(x.x), fabsf(x.y)};
  const gelu_f32x2 z = ax * 0.70710678118654752440f;
  const gelu_f32x2 den = __builtin_elementwise_fma(gelu_f32x2{0.3275911f, 0.3275911f}, z, gelu_f32x2{1.f, 1.f});
  const gelu_f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  gelu_f32x2 poly = __builtin_elementwise_fma(t, gelu_f32x2{1.061405429f, 1.061405429f}, gelu_f32x2{-1.453152027f, -1.453152027f});
  poly = __builtin_elementwise_fma(t, poly, gelu_f32x2{1.421413741f, 1.421413741f});
  poly = __builtin_elementwise_fma(t, poly, gelu_f32x2{-0.284496736f, -0.284496736f});
  poly = __builtin_elementwise_fma(t, poly, gelu_f32x2{0.254829592f, 0.254829592f});
  const gelu_f32x2 arg = (x * x) * -0.72134752044448170368f;                // e^{-x^2 / 2} = 2^arg
  e = gelu_f32x2{__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
  const gelu_f32x2 q = ((poly * t) * 0.5f) * e;                             // (1 - erf(z)) / 2
  const gelu_f32x2 omq = gelu_f32x2{1.f, 1.f} - q;
  cdf = gelu_f32x2{x.x >= 0.f ? omq.x : q.x, x.y >= 0.f ? omq.y : q.y};
}
__device__ __forceinline__ gelu_f32x2 gelu_f2(gelu_f32x2 x) {
#pragma clang fp contract(off)
  gelu_f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  return x * cdf;
}
// d gelu(x) / dx * dy = (cdf + x pdf) dy,  pdf = e^{-x^2 / 2} / sqrt(2 pi)
__device__ __forceinline__ gelu_f32x2 gelu_grad_f2(gelu_f32x2 x, gelu_f32x2 dy) {
#pragma clang fp contract(off)
  gelu_f32x2 cdf, e;
  gelu_parts2(x, cdf, e);
  return dy * __builtin_elementwise_fma(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float gelu_f(float x) { return gelu_f2(gelu_f32x2{x, 0.f}).x; }
__device__ __forceinline__ float gelu_grad_f(float x, float dy) { return gelu_grad_f2(gelu_f32x2{x, 0.f}, gelu_f32x2{dy, 0.f}).x; }

// RoPE rotation of one (x[i], x[i + D/2]) pair by (cos, sin) — ONE definition (explicit fma order) for the row kernel
// (norm_act.hip rope_apply_kernel) and the GEMM's RoPE epilogue (gemm.hip EPI_ROPE), which must agree bit for bit
__device__ __forceinline__ void rope_rotate(float a, float b, float c, float s, float& ya, float& yb) {
  ya = __builtin_fmaf(-b, s, a * c);
  yb = __builtin_fmaf(a, s, b * c);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte vector of T: 8 bf16 or 4 fp32 -----------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  uint4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ __forceinline__ void unpack(float (&f)[8]) const {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(w[i] << 16);
      f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void pack(const float (&f)[8]) {
    raw.x = pack2bf(f[0], f[1]);
    raw.y = pack2bf(f[2], f[3]);
    raw.z = pack2bf(f[4], f[5]);
    raw.w = pack2bf(f[6], f[7]);
  }
};
template <> struct Vec16<float> {
  static constexpr int N = 4;
  float4 raw;
  __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
  __device__ __forceinline__ void unpack(float (&f)[4]) const {
    f[0] = raw.x; f[1] = raw.y; f[2] = raw.z; f[3] = raw.w;
  }
  __device__ __forceinline__ void pack(const float (&f)[4]) { raw = make_float4(f[0], f[1], f[2], f[3]); }
};

// wave64 / block reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum; `sm` must hold >= blockDim.x/64 floats; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sm[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float r = sm[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, sm[i]);
  return r;
}

}  // namespace tn
