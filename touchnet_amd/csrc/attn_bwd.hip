// Packed (document-masked, causal) flash attention BACKWARD for gfx950 — MFMA 32x32x16 bf16.
//
// Same masking / layout contract as attn_fwd.hip.  Three kernels, all deterministic (no atomics):
//   1. delta[b,h,t] = sum_d dO[b,t,h,d] * O[b,t,h,d]                                (HBM-bound)
//   2. dK/dV: a workgroup owns 128 KV rows of one KV head (32 per wave, K and V held in registers as
//      MFMA B operands) and walks the query tiles (32 rows) of every query head of its GQA group that
//      can see those rows:   S = Q K^T, P = exp2(S*c - LSE2), dP = dO V^T, dS = P o (dP - delta),
//      dV^T += dO^T P, dK^T += Q^T dS.  S is computed un-transposed here so that the contraction index
//      of the last two products (q) is the in-lane index of P / dS.
//   3. dQ: a workgroup owns 128 query rows of one head and walks KV tiles exactly like the forward:
//      S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T.
// Recomputing S in both (7 matmuls instead of 5) buys determinism and needs no fp32 dQ scratch; the
// reference's flex_attention backward does the same (inductor-generated two-loop Triton template).
#include "attn_common.h"

namespace tn {

// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, const bf16_t* __restrict__ dO,
                                                         float* __restrict__ delta, int B, int T, int Nh) {
  constexpr int LPR = D / 8;        // lanes per (token, head) row
  const size_t rows = (size_t)B * T * Nh;
  const size_t row = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
  const int sub = threadIdx.x % LPR;
  float acc = 0.f;
  if (row < rows) {
    Vec16<bf16_t> a, g;
    float af[8], gf[8];
    a.load(O + row * D + sub * 8);
    g.load(dO + row * D + sub * 8);
    a.unpack(af);
    g.unpack(gf);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += af[j] * gf[j];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (row < rows && sub == 0) {
    const size_t h = row % Nh, t = (row / Nh) % T, b = row / ((size_t)Nh * T);
    delta[(b * Nh + h) * T + t] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// dK / dV
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ Delta,
    bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, const int* __restrict__ doc, AttnMeta meta, int T, int Nh,
    int Nkv, float scale, float scale_log2) {
  constexpr int BNK = 128, BQ = 32;
  constexpr int KSTEPS = D / 16, DBLK = D / 32, LD = D + 8, TS = TLds<BQ>::STRIDE;
  // LDS: Qs | dOs (row-major [32][D+8]) | Qt | dOt (transposed [D][48]) | lse[32] delta[32] docq[32]
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BQ * LD + 2 * D * TS + 6 * BQ];
  bf16_t* Qs = smem;
  bf16_t* dOs = Qs + BQ * LD;
  bf16_t* Qt = dOs + BQ * LD;
  bf16_t* dOt = Qt + D * TS;
  float* lse_s = reinterpret_cast<float*>(dOt + D * TS);
  float* delta_s = lse_s + BQ;
  int* docq = reinterpret_cast<int*>(delta_s + BQ);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int kt = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
  const int G = Nh / Nkv;
  const int k0 = kt * BNK;
  const int wk0 = k0 + 32 * wave;
  const int kvrow = wk0 + l31;
  const bool kvalid = kvrow < T;

  bf16x8_t kreg[KSTEPS], vreg[KSTEPS];
  {
    const size_t off = (((size_t)b * T + (kvalid ? kvrow : 0)) * Nkv + hk) * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
      if (kvalid) {
        a = *reinterpret_cast<const uint4*>(K + off + 16 * s);
        c = *reinterpret_cast<const uint4*>(V + off + 16 * s);
      }
      kreg[s] = as_bf16x8(a);
      vreg[s] = as_bf16x8(c);
    }
  }
  const int dkdoc = kvalid ? doc[(size_t)b * T + kvrow] : 0;
  int wminpos = dkdoc > 0 ? dkdoc : 0x7fffffff, wmax = dkdoc;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    wminpos = min(wminpos, __shfl_xor(wminpos, o, 64));
    wmax = max(wmax, __shfl_xor(wmax, o, 64));
  }

  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = 2 * kt, t1 = min(2 * kt + 1, meta.nt - 1);
  const int bminpos = min(m_minpos[t0], m_minpos[t1]);
  const int bmax = max(m_max[t0], m_max[t1]);
  const int qhi64 = max(meta.kv_hi[(size_t)b * meta.nt + t0], meta.kv_hi[(size_t)b * meta.nt + t1]);
  const int qt_lo = k0 / BQ;                                        // first 32-row query tile (q >= kv)
  const int qt_end = min((qhi64 + 1) * (kTile / BQ), (T + BQ - 1) / BQ);  // exclusive
  const int nqt = max(qt_end - qt_lo, 0);
  const int n_it = nqt * G;                                         // flattened (head-in-group, q tile)
  auto advance = [&](int it) {
    while (it < n_it) {
      const int t64 = (qt_lo + it % nqt) * BQ / kTile;
      if (tile_may_interact(m_minpos[t64], m_max[t64], bminpos, bmax)) break;
      ++it;
    }
    return it;
  };

  f32x16_t dkacc[DBLK], dvacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dkacc[i][r] = dvacc[i][r] = 0.f;

  TransposeStage<BQ, D, 256> qst, dost;
  float lse_st = 0.f, delta_st = 0.f;
  int doc_st = 0;
  const size_t qld = (size_t)Nh * D;
  auto issue = [&](int it) {
    const int g = it / nqt, qb = (qt_lo + it % nqt) * BQ;
    const int h = hk * G + g;
    const size_t base = (((size_t)b * T + qb) * Nh + h) * D;
    qst.load(Q + base, qld, T - qb, tid);
    dost.load(dO + base, qld, T - qb, tid);
    if (tid < BQ) {
      const bool ok = qb + tid < T;
      const size_t si = ((size_t)b * Nh + h) * T + qb + tid;
      lse_st = ok ? LSE2[si] : INFINITY;
      delta_st = ok ? Delta[si] : 0.f;
      doc_st = ok ? doc[(size_t)b * T + qb + tid] : 0;
    }
  };
  // write both images (row-major and transposed) of a staged 32 x D tile
  auto stage_store = [&](const TransposeStage<BQ, D, 256>& st, bf16_t* rm, bf16_t* tr) {
    constexpr int CPR = D / 8;
    if (tid < (BQ / 4) * CPR) {
      const int r4 = tid / CPR, c8 = tid % CPR;
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(rm + (4 * r4 + k) * LD + c8 * 8) = st.v[0][k];
    }
    st.store(tr, tid);
  };

  int it = advance(0);
  if (it < n_it) issue(it);
  while (it < n_it) {
    const int itn = advance(it + 1);
    __syncthreads();
    stage_store(qst, Qs, Qt);
    stage_store(dost, dOs, dOt);
    if (tid < BQ) {
      lse_s[tid] = lse_st;
      delta_s[tid] = delta_st;
      docq[tid] = doc_st;
    }
    __syncthreads();
    if (itn < n_it) issue(itn);

    const int qb = (qt_lo + it % nqt) * BQ;
    const int t64 = qb / kTile;
    if (qb + BQ - 1 >= wk0 && tile_may_interact(m_minpos[t64], m_max[t64], wminpos, wmax)) {
      // ---- S[q, kv] = Q K^T ; dP[q, kv] = dO V^T     (rows = q in registers, column = this lane's kv)
      f32x16_t sacc, dpacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = dpacc[r] = 0.f;
      const bf16_t* qp = Qs + l31 * LD + 8 * hi;
      const bf16_t* dop = dOs + l31 * LD + 8 * hi;
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        sacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(qp + 16 * s)), kreg[s], sacc);
        dpacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(dop + 16 * s)), vreg[s], dpacc);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 8 * r4 + 4 * hi);
        const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 8 * r4 + 4 * hi);
        const int4 q4 = *reinterpret_cast<const int4*>(docq + 8 * r4 + 4 * hi);
        const float le[4] = {l4.x, l4.y, l4.z, l4.w}, de[4] = {d4.x, d4.y, d4.z, d4.w};
        const int qd[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * r4 + e;
          const int qi = qb + 8 * r4 + 4 * hi + e;
          const bool ok = (kvrow <= qi) && (qd[e] == dkdoc) && (dkdoc > 0);
          const float pv = ok ? fast_exp2(sacc[r] * scale_log2 - le[e]) : 0.f;
          p[r] = pv;
          ds[r] = pv * (dpacc[r] - de[e]);
        }
      }
      bf16x8_t pb[2], dsb[2];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        u32x4_t t = {pack2bf(p[8 * sp + 0], p[8 * sp + 1]), pack2bf(p[8 * sp + 2], p[8 * sp + 3]),
                     pack2bf(p[8 * sp + 4], p[8 * sp + 5]), pack2bf(p[8 * sp + 6], p[8 * sp + 7])};
        pb[sp] = __builtin_bit_cast(bf16x8_t, t);
        u32x4_t u = {pack2bf(ds[8 * sp + 0], ds[8 * sp + 1]), pack2bf(ds[8 * sp + 2], ds[8 * sp + 3]),
                     pack2bf(ds[8 * sp + 4], ds[8 * sp + 5]), pack2bf(ds[8 * sp + 6], ds[8 * sp + 7])};
        dsb[sp] = __builtin_bit_cast(bf16x8_t, u);
      }
      // ---- dV^T[d, kv] += dO^T[d, q] P[q, kv] ;  dK^T[d, kv] += Q^T[d, q] dS[q, kv]
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        const int d = 32 * db + l31;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const int g0 = 4 * sp + hi;
          const uint2 a0 = *reinterpret_cast<const uint2*>(dOt + TLds<BQ>::off(d, g0));
          const uint2 a1 = *reinterpret_cast<const uint2*>(dOt + TLds<BQ>::off(d, g0 + 2));
          dvacc[db] = mfma32(as_bf16x8(a0, a1), pb[sp], dvacc[db]);
          const uint2 c0 = *reinterpret_cast<const uint2*>(Qt + TLds<BQ>::off(d, g0));
          const uint2 c1 = *reinterpret_cast<const uint2*>(Qt + TLds<BQ>::off(d, g0 + 2));
          dkacc[db] = mfma32(as_bf16x8(c0, c1), dsb[sp], dkacc[db]);
        }
      }
    }
    it = itn;
  }

  if (kvalid) {
    const size_t off = (((size_t)b * T + kvrow) * Nkv + hk) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 o;
        o.x = pack2bf(dkacc[db][4 * r4 + 0] * scale, dkacc[db][4 * r4 + 1] * scale);
        o.y = pack2bf(dkacc[db][4 * r4 + 2] * scale, dkacc[db][4 * r4 + 3] * scale);
        *reinterpret_cast<uint2*>(dK + off + 32 * db + 8 * r4 + 4 * hi) = o;
        o.x = pack2bf(dvacc[db][4 * r4 + 0], dvacc[db][4 * r4 + 1]);
        o.y = pack2bf(dvacc[db][4 * r4 + 2], dvacc[db][4 * r4 + 3]);
        *reinterpret_cast<uint2*>(dV + off + 32 * db + 8 * r4 + 4 * hi) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dQ
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ Delta,
    bf16_t* __restrict__ dQ, const int* __restrict__ doc, AttnMeta meta, int T, int Nh, int Nkv, float scale,
    float scale_log2) {
  constexpr int BM = 128, BN = 64;
  constexpr int KSTEPS = D / 16, DBLK = D / 32, LD = D + 8, TS = TLds<BN>::STRIDE;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BN * LD + D * TS + 2 * BN];
  bf16_t* Ks = smem;
  bf16_t* Vs = Ks + BN * LD;
  bf16_t* Kt = Vs + BN * LD;
  int* docs = reinterpret_cast<int*>(Kt + D * TS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hk = h / (Nh / Nkv);
  const int q0 = qt * BM;
  const int wq0 = q0 + 32 * wave;
  const int qrow = wq0 + l31;
  const bool qvalid = qrow < T;

  bf16x8_t qreg[KSTEPS], doreg[KSTEPS];
  {
    const size_t off = (((size_t)b * T + (qvalid ? qrow : 0)) * Nh + h) * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
      if (qvalid) {
        a = *reinterpret_cast<const uint4*>(Q + off + 16 * s);
        c = *reinterpret_cast<const uint4*>(dO + off + 16 * s);
      }
      qreg[s] = as_bf16x8(a);
      doreg[s] = as_bf16x8(c);
    }
  }
  const int dq = qvalid ? doc[(size_t)b * T + qrow] : 0;
  const float lse2 = qvalid ? LSE2[((size_t)b * Nh + h) * T + qrow] : INFINITY;
  const float delta = qvalid ? Delta[((size_t)b * Nh + h) * T + qrow] : 0.f;
  int wminpos = dq > 0 ? dq : 0x7fffffff, wmax = dq;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    wminpos = min(wminpos, __shfl_xor(wminpos, o, 64));
    wmax = max(wmax, __shfl_xor(wmax, o, 64));
  }

  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = 2 * qt, t1 = min(2 * qt + 1, meta.nt - 1);
  const int bminpos = min(m_minpos[t0], m_minpos[t1]);
  const int bmax = max(m_max[t0], m_max[t1]);
  const int j_hi = t1;
  int j = min(meta.q_lo[(size_t)b * meta.nt + t0], meta.q_lo[(size_t)b * meta.nt + t1]);
  auto advance = [&](int jj) {
    while (jj <= j_hi && !tile_may_interact(bminpos, bmax, m_minpos[jj], m_max[jj])) ++jj;
    return jj;
  };
  j = advance(j);

  f32x16_t dqacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  TransposeStage<BN, D, 256> kst;   // K: written row-major AND transposed from the same registers
  RowMajorStage<BN, D, 256> vst;
  int dstage = 0;
  const size_t kvld = (size_t)Nkv * D;
  auto issue = [&](int jj) {
    const int k0 = jj * BN;
    const size_t base = (((size_t)b * T + k0) * Nkv + hk) * D;
    kst.load(K + base, kvld, T - k0, tid);
    vst.load(V + base, kvld, T - k0, tid);
    if (tid < BN) dstage = (k0 + tid < T) ? doc[(size_t)b * T + k0 + tid] : 0;
  };
  if (j <= j_hi) issue(j);

  while (j <= j_hi) {
    const int jn = advance(j + 1);
    __syncthreads();
    {
      constexpr int CPR = D / 8, UNITS = (BN / 4) * CPR;
#pragma unroll
      for (int i = 0; i < TransposeStage<BN, D, 256>::N; ++i) {
        const int u = tid + i * 256;
        if (u < UNITS) {
          const int r4 = u / CPR, c8 = u % CPR;
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(Ks + (4 * r4 + k) * LD + c8 * 8) = kst.v[i][k];
        }
      }
      kst.store(Kt, tid);
      vst.store(Vs, tid);
      if (tid < BN) docs[tid] = dstage;
    }
    __syncthreads();
    if (jn <= j_hi) issue(jn);

    const int k0 = j * BN;
    if (k0 <= wq0 + 31 && tile_may_interact(wminpos, wmax, m_minpos[j], m_max[j])) {
      f32x16_t sacc[2], dpacc[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[blk][r] = dpacc[blk][r] = 0.f;
        const bf16_t* kp = Ks + (32 * blk + l31) * LD + 8 * hi;
        const bf16_t* vp = Vs + (32 * blk + l31) * LD + 8 * hi;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          sacc[blk] = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(kp + 16 * s)), qreg[s], sacc[blk]);
          dpacc[blk] = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(vp + 16 * s)), doreg[s], dpacc[blk]);
        }
      }
      bf16x8_t dsb[2][2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        float ds[16];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int4 dk = *reinterpret_cast<const int4*>(docs + 32 * blk + 8 * r4 + 4 * hi);
          const int dkk[4] = {dk.x, dk.y, dk.z, dk.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * r4 + e;
            const int kv = k0 + 32 * blk + 8 * r4 + 4 * hi + e;
            const bool ok = (kv <= qrow) && (dkk[e] == dq) && (dq > 0);
            const float pv = ok ? fast_exp2(sacc[blk][r] * scale_log2 - lse2) : 0.f;
            ds[r] = pv * (dpacc[blk][r] - delta);
          }
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          u32x4_t u = {pack2bf(ds[8 * sp + 0], ds[8 * sp + 1]), pack2bf(ds[8 * sp + 2], ds[8 * sp + 3]),
                       pack2bf(ds[8 * sp + 4], ds[8 * sp + 5]), pack2bf(ds[8 * sp + 6], ds[8 * sp + 7])};
          dsb[blk][sp] = __builtin_bit_cast(bf16x8_t, u);
        }
      }
      // ---- dQ^T[d, q] += K^T[d, kv] dS^T[kv, q]
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
        const int d = 32 * db + l31;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            const int g0 = 8 * blk + 4 * sp + hi;
            const uint2 a0 = *reinterpret_cast<const uint2*>(Kt + TLds<BN>::off(d, g0));
            const uint2 a1 = *reinterpret_cast<const uint2*>(Kt + TLds<BN>::off(d, g0 + 2));
            dqacc[db] = mfma32(as_bf16x8(a0, a1), dsb[blk][sp], dqacc[db]);
          }
        }
      }
    }
    j = jn;
  }

  if (qvalid) {
    bf16_t* op = dQ + (((size_t)b * T + qrow) * Nh + h) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 o;
        o.x = pack2bf(dqacc[db][4 * r4 + 0] * scale, dqacc[db][4 * r4 + 1] * scale);
        o.y = pack2bf(dqacc[db][4 * r4 + 2] * scale, dqacc[db][4 * r4 + 3] * scale);
        *reinterpret_cast<uint2*>(op + 32 * db + 8 * r4 + 4 * hi) = o;
      }
    }
  }
}

}  // namespace tn

using namespace tn;

extern "C" {

// delta: float [B, Nh, T] scratch (also an output of this call).  dq/dk/dv are fully overwritten.
int tn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                int Nkv, int D, float scale, void* stream) {
  if (B <= 0 || T <= 0 || Nh <= 0 || Nkv <= 0 || Nh % Nkv) return TN_EINVAL;
  if (D != 64 && D != 128) return TN_EINVAL;
  const int nt = (T + kTile - 1) / kTile, n = B * nt;
  AttnMeta m = {meta, meta + n, meta + 2 * n, meta + 3 * n, meta + 4 * n, nt};
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  const size_t rows = (size_t)B * T * Nh;
  dim3 gq((T + 127) / 128, Nh, B), gk((T + 127) / 128, Nkv, B), block(256);
  if (D == 128) {
    hipLaunchKernelGGL((attn_delta_kernel<128>), dim3((rows * 16 + 255) / 256), block, 0, st, (const bf16_t*)o,
                       (const bf16_t*)dout, delta, B, T, Nh);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128>), gk, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, lse2, delta, (bf16_t*)dk, (bf16_t*)dv, doc, m, T, Nh,
                       Nkv, scale, sl2);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<128>), gq, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, lse2, delta, (bf16_t*)dq, doc, m, T, Nh, Nkv, scale,
                       sl2);
  } else {
    hipLaunchKernelGGL((attn_delta_kernel<64>), dim3((rows * 8 + 255) / 256), block, 0, st, (const bf16_t*)o,
                       (const bf16_t*)dout, delta, B, T, Nh);
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64>), gk, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, lse2, delta, (bf16_t*)dk, (bf16_t*)dv, doc, m, T, Nh,
                       Nkv, scale, sl2);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), gq, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (const bf16_t*)dout, lse2, delta, (bf16_t*)dq, doc, m, T, Nh, Nkv, scale,
                       sl2);
  }
  TN_LAUNCH_CHECK();
  return TN_OK;
}

}  // extern "C"
