// BEST-RQ label generation on the device: random projection + nearest normalised codebook entry.
//
// Replaces BestRQTokenizer.tokenize (touchnet/tokenizer/tokenizer.py:289-299), which the reference runs per
// utterance on CPU dataloader workers and materialises a [T, V, E] difference tensor for:
//     x = feat @ quantizer;  x /= max(||x||, 1e-8);  code = argmin_v || x - codebook[v] ||_2   (first index on ties)
// fp32 throughout and the same formulation (explicit differences, sqrt, strict-less argmin in ascending v), so the
// codes agree with the reference wherever its own best/second-best margin exceeds fp32 round-off.
//
// Not GEMM-shaped enough for MFMA at fp32 (E = 16): VALU-bound.  One 256-thread workgroup owns FR = 32 frames:
//   1. projection: feature tile and projection chunk staged through LDS (coalesced 16-byte loads), each thread
//      accumulates FR*E/256 outputs in ascending-k fma order;
//   2. normalisation by one thread per frame;
//   3. every thread scans codebook rows v = tid, tid+256, ... (L2-resident) against the frames, 128/E frames at a
//      time with their x vectors in registers, keeping a running (distance, index) per frame;
//   4. lexicographic (distance, index) min over the wave by shuffles, over the 4 waves through LDS.
#include "common.h"

namespace tn {

template <int E>
__global__ __launch_bounds__(256) void bestrq_tokenize_kernel(const float* __restrict__ feat,
                                                              const float* __restrict__ quantizer,
                                                              const float* __restrict__ codebook,
                                                              long long* __restrict__ codes, int T, int F, int V) {
  constexpr int FR = 32, KC = 64, PER = FR * E / 256;       // outputs per thread in the projection
  static_assert(FR * E % 256 == 0 && E % 4 == 0, "unsupported embedding size");
  __shared__ float ftile[FR][KC + 1];
  __shared__ __attribute__((aligned(16))) float qtile[KC][E];
  __shared__ __attribute__((aligned(16))) float xs[FR][E];
  __shared__ float wbest[4][FR];
  __shared__ int widx[4][FR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f0 = blockIdx.x * FR;

  // ---- 1. projection
  float acc[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < F; k0 += KC) {
    for (int i = tid; i < FR * KC; i += 256) {
      const int f = i / KC, k = i % KC;
      ftile[f][k] = (f0 + f < T && k0 + k < F) ? feat[(size_t)(f0 + f) * F + k0 + k] : 0.f;
    }
    for (int i = tid; i < KC * E; i += 256) {
      const int k = i / E, e = i % E;
      qtile[k][e] = (k0 + k < F) ? quantizer[(size_t)(k0 + k) * E + e] : 0.f;
    }
    __syncthreads();
    const int kn = min(KC, F - k0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int o = tid + 256 * i, f = o / E, e = o % E;
      float a = acc[i];
      for (int k = 0; k < kn; ++k) a = fmaf(ftile[f][k], qtile[k][e], a);
      acc[i] = a;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int o = tid + 256 * i;
    xs[o / E][o % E] = acc[i];
  }
  __syncthreads();
  // ---- 2. L2 normalisation (F.normalize, eps 1e-8)
  if (tid < FR) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s = fmaf(xs[tid][e], xs[tid][e], s);
    const float n = fmaxf(sqrtf(s), 1e-8f);
#pragma unroll
    for (int e = 0; e < E; ++e) xs[tid][e] = xs[tid][e] / n;
  }
  __syncthreads();

  // ---- 3. nearest codebook row per frame
  float best[FR];
  int bidx[FR];
#pragma unroll
  for (int f = 0; f < FR; ++f) {
    best[f] = INFINITY;
    bidx[f] = 0x7fffffff;
  }
  // frames in chunks of FC whose x vectors sit in registers (128 floats) while the thread walks its codebook rows
  // (left to itself hipcc hoists ALL FR*E loop-invariant LDS reads out of the v loop and spills)
  constexpr int FC = 128 / E;
#pragma unroll
  for (int fc = 0; fc < FR; fc += FC) {
    float xr[FC][E];
#pragma unroll
    for (int f = 0; f < FC; ++f)
#pragma unroll
      for (int e4 = 0; e4 < E / 4; ++e4) {
        const float4 x = *reinterpret_cast<const float4*>(&xs[fc + f][4 * e4]);   // same address in every lane
        xr[f][4 * e4 + 0] = x.x; xr[f][4 * e4 + 1] = x.y; xr[f][4 * e4 + 2] = x.z; xr[f][4 * e4 + 3] = x.w;
      }
    for (int v = tid; v < V; v += 256) {
      float c[E];
#pragma unroll
      for (int e4 = 0; e4 < E / 4; ++e4) {
        const float4 t = reinterpret_cast<const float4*>(codebook + (size_t)v * E)[e4];
        c[4 * e4 + 0] = t.x; c[4 * e4 + 1] = t.y; c[4 * e4 + 2] = t.z; c[4 * e4 + 3] = t.w;
      }
#pragma unroll
      for (int f = 0; f < FC; ++f) {
        float d2 = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const float df = xr[f][e] - c[e];
          d2 = fmaf(df, df, d2);
        }
        const float d = sqrtf(d2);
        if (d < best[fc + f]) {      // strict: the smallest v of this thread wins ties (v ascends)
          best[fc + f] = d;
          bidx[fc + f] = v;
        }
      }
    }
  }
  // ---- 4. (distance, index) lexicographic min: wave shuffles, then the 4 waves
#pragma unroll
  for (int f = 0; f < FR; ++f) {
    float d = best[f];
    int ix = bidx[f];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(d, o, 64);
      const int oi = __shfl_xor(ix, o, 64);
      if (od < d || (od == d && oi < ix)) {
        d = od;
        ix = oi;
      }
    }
    if (lane == 0) {
      wbest[wave][f] = d;
      widx[wave][f] = ix;
    }
  }
  __syncthreads();
  if (tid < FR && f0 + tid < T) {
    float d = wbest[0][tid];
    int ix = widx[0][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float od = wbest[w][tid];
      const int oi = widx[w][tid];
      if (od < d || (od == d && oi < ix)) {
        d = od;
        ix = oi;
      }
    }
    codes[f0 + tid] = ix;
  }
}

}  // namespace tn

extern "C" int tn_bestrq_tokenize(const float* feat, const float* quantizer, const float* codebook, long long* codes,
                                  int T, int F, int E, int V, void* stream) {
  if (T <= 0) return TN_OK;
  if (F <= 0 || V <= 0 || ((uintptr_t)codebook & 15)) return TN_EINVAL;
  dim3 grid((T + 31) / 32), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (E) {
    case 8: hipLaunchKernelGGL((tn::bestrq_tokenize_kernel<8>), grid, block, 0, st, feat, quantizer, codebook, codes, T, F, V); break;
    case 16: hipLaunchKernelGGL((tn::bestrq_tokenize_kernel<16>), grid, block, 0, st, feat, quantizer, codebook, codes, T, F, V); break;
    case 32: hipLaunchKernelGGL((tn::bestrq_tokenize_kernel<32>), grid, block, 0, st, feat, quantizer, codebook, codes, T, F, V); break;
    default: return TN_EINVAL;
  }
  TN_LAUNCH_CHECK();
  return TN_OK;
}
