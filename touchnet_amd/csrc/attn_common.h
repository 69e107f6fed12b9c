// Shared pieces of the packed (document-masked, causal) attention kernels for gfx950.
//
// Mask predicate (bit-exact restatement of transformers/integrations/flex_attention.py:190-201 as
// invoked at touchnet/models/kimi_audio/modeling_kimi_audio.py:582-585 for the packers' document ids,
// touchnet/models/llama/processing_llama.py:38-40):
//     allow(b, q, kv) = (q >= kv) && (doc[b,q] > 0) && (doc[b,q] == doc[b,kv])
// Rows with no allowed key (pad rows, doc == 0) produce 0 output / 0 gradient (flex semantics).
//
// MFMA conventions used everywhere (v_mfma_f32_32x32x16_bf16, wave64):
//   operand A[i][k] / B[k][j]: lane holds i (resp. j) = lane & 31 and 8 contraction slots selected by
//     (lane >> 5, e) — the same slot function for A and B, so any consistent choice of which logical
//     index feeds slot (hi, e) is valid;
//   result  C[i][j]: lane holds column j = lane & 31, register r holds row
//     i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma once
#include "common.h"

namespace tn {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t as_bf16x8(uint4 v) {
  u32x4_t t = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(bf16x8_t, t);
}
__device__ __forceinline__ bf16x8_t as_bf16x8(uint2 lo, uint2 hi) {
  u32x4_t t = {lo.x, lo.y, hi.x, hi.y};
  return __builtin_bit_cast(bf16x8_t, t);
}
// row index inside a 32x32 MFMA result tile held by register r of a lane with hi = lane >> 5
__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

constexpr int kTile = 64;  // granularity of the document-range metadata

// Per-64-position tile metadata (all int32 [B, nt], nt = ceil(T/64)):
//   tmax    max document id in the tile (positions >= T count as 0)
//   tminpos min over the POSITIVE ids of the tile (INT_MAX if the tile is all pad)
//   tmin    min over all ids (0 if the tile contains pad)
//   q_lo    first kv tile j <= t that can interact with q tile t        (t + 1 if none)
//   kv_hi   last  q  tile t >= j that can interact with kv tile j       (j - 1 if none)
struct AttnMeta {
  const int* tmin;
  const int* tmax;
  const int* tminpos;
  const int* q_lo;
  const int* kv_hi;
  int nt;
};

// Can any (q, kv) pair with q in a set having positive-id range [qminpos, qmax] and kv in tile j be
// allowed?  Conservative on purpose: false only when the id ranges are disjoint.
__device__ __forceinline__ bool tile_may_interact(int qminpos, int qmax, int kminpos, int kmax) {
  return !(qminpos == 0x7fffffff || kminpos == 0x7fffffff || kmax < qminpos || kminpos > qmax);
}

// Transposed LDS image [D rows][R source rows] of an [R][D] bf16 tile: element (d, s) lives at
//   d * STRIDE + 4 * ((s >> 2) ^ swz(d)) + (s & 3),  swz(d) = (d >> 3) & (R/4 - 1)
// Written as 8-byte (4 source rows) groups, read as 8-byte groups by the MFMA operand loads:
// conflict-free for both on gfx950 with STRIDE = R + 16 (simulated against the LDS bank rules of
// MI355X_MICROARCH.md).
template <int R>
struct TLds {
  static constexpr int STRIDE = R + 16;
  static __device__ __forceinline__ int off(int d, int g) { return d * STRIDE + 4 * ((g ^ (d >> 3)) & (R / 4 - 1)); }
};

// Stage a [R rows][D] bf16 tile, source row stride `ld` elements, rows >= rows_valid zero-filled.
//  * row-major image  dst_rm[r * (D + 8) + d]          (16-byte writes)
//  * transposed image dst_t  via TLds<R>               (8-byte writes of 4 consecutive source rows)
// Split in two phases so the global loads can be issued early and the LDS writes late.
template <int R, int D, int NT>
struct RowMajorStage {
  static constexpr int CPR = D / 8;                       // 16-byte chunks per row
  static constexpr int N = (R * CPR + NT - 1) / NT;       // chunks per thread
  uint4 v[N];
  __device__ __forceinline__ void load(const bf16_t* src, size_t ld, int rows_valid, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + i * NT;
      const int r = c / CPR, cc = c % CPR;
      v[i] = make_uint4(0, 0, 0, 0);
      if (c < R * CPR && r < rows_valid) v[i] = *reinterpret_cast<const uint4*>(src + (size_t)r * ld + cc * 8);
    }
  }
  __device__ __forceinline__ void store(bf16_t* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + i * NT;
      const int r = c / CPR, cc = c % CPR;
      if (c < R * CPR) *reinterpret_cast<uint4*>(dst + r * (D + 8) + cc * 8) = v[i];
    }
  }
};

template <int R, int D, int NT>
struct TransposeStage {
  static constexpr int CPR = D / 8;
  static constexpr int UNITS = (R / 4) * CPR;             // one unit = 4 rows x 8 columns
  static constexpr int N = (UNITS + NT - 1) / NT;
  uint4 v[N][4];
  __device__ __forceinline__ void load(const bf16_t* src, size_t ld, int rows_valid, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int u = tid + i * NT;
      const int r4 = u / CPR, c8 = u % CPR;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = 4 * r4 + k;
        v[i][k] = make_uint4(0, 0, 0, 0);
        if (u < UNITS && r < rows_valid) v[i][k] = *reinterpret_cast<const uint4*>(src + (size_t)r * ld + c8 * 8);
      }
    }
  }
  __device__ __forceinline__ void store(bf16_t* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int u = tid + i * NT;
      const int r4 = u / CPR, c8 = u % CPR;
      if (u < UNITS) {
        const uint32_t w[4][4] = {{v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w},
                                  {v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w},
                                  {v[i][2].x, v[i][2].y, v[i][2].z, v[i][2].w},
                                  {v[i][3].x, v[i][3].y, v[i][3].z, v[i][3].w}};
#pragma unroll
        for (int dd = 0; dd < 8; ++dd) {
          const int wi = dd >> 1;
          uint2 o;
          if (dd & 1) {
            o.x = (w[0][wi] >> 16) | (w[1][wi] & 0xffff0000u);
            o.y = (w[2][wi] >> 16) | (w[3][wi] & 0xffff0000u);
          } else {
            o.x = (w[0][wi] & 0xffffu) | (w[1][wi] << 16);
            o.y = (w[2][wi] & 0xffffu) | (w[3][wi] << 16);
          }
          *reinterpret_cast<uint2*>(dst + TLds<R>::off(c8 * 8 + dd, r4)) = o;
        }
      }
    }
  }
};

}  // namespace tn
