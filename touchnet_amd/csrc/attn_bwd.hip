// Packed (document-masked, causal) flash attention BACKWARD for gfx950 — MFMA 32x32x16 bf16.
//
// Same masking / layout contract as attn_fwd.hip.  All kernels are deterministic (no atomics):
//   1. delta[b,h,t] = sum_d dO[b,t,h,d] * O[b,t,h,d]                                (HBM-bound)
//   2. dK / dV: a workgroup owns 128 KV rows of one KV head (32 per wave, K/V rows held in registers as MFMA
//      B operands) and walks the 64-row query tiles of every query head of its GQA group that can see them:
//          S = Q K^T,  P = exp2(S*c - LSE2),  dP = dO V^T,  dS = P o (dP - delta),
//          dV^T += dO^T P,   dK^T += Q^T dS
//      S is computed un-transposed here so that the contraction index of the last two products (q) is the
//      in-lane index of P / dS.  MODE selects what one launch accumulates:
//        D = 64 : one kernel does both (2 waves/SIMD)
//        D = 128: a dV launch and a dK launch.  Keeping dK and dV accumulators (128 regs) plus K and V
//                 operands (64) resident needs the 512-register file at 1 wave/SIMD and made hipcc shuttle
//                 ~14 v_accvgpr moves per MFMA (measured: 15 VALU per MFMA, SQ_INSTS_VALU / SQ_INSTS_MFMA);
//                 two lean kernels at 2 waves/SIMD recompute S once more (+25 % MFMA) and are faster.
//   3. dQ: a workgroup owns 128 query rows of one head and walks KV tiles exactly like the forward:
//      S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T, one 32-row KV block at a time.
// Recomputing S in each pass buys determinism and needs no fp32 dQ scratch; the reference's
// flex_attention backward has the same two-loop structure (inductor-generated Triton template).
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
// dK / dV.   MODE 0: dV only   1: dK only   2: both
// ------------------------------------------------------------------------------------------------
template <int D, int MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ Delta,
    bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, const int* __restrict__ doc, AttnMeta meta, QView qv, int T,
    int Nh, int Nkv, float scale, float scale_log2) {
  constexpr bool DO_DV = MODE != 1, DO_DK = MODE != 0;
  constexpr int BNK = 128, BQ = 64;
  constexpr int KSTEPS = D / 16, DBLK = D / 32, LD = D + 8, TS = TLds<BQ>::STRIDE;
  // LDS images of the current 64-row query tile:
  //   Qs  row-major  (always: A operand of S = Q K^T)      dOs row-major  (dK: A operand of dP = dO V^T)
  //   Qt  transposed (dK: A operand of dK^T += Q^T dS)     dOt transposed (dV: A operand of dV^T += dO^T P)
  constexpr int N_RM = DO_DK ? 2 : 1, N_TR = (DO_DK ? 1 : 0) + (DO_DV ? 1 : 0);
  __shared__ __attribute__((aligned(16))) bf16_t smem[N_RM * BQ * LD + N_TR * D * TS + 6 * BQ];
  bf16_t* Qs = smem;
  bf16_t* dOs = Qs + BQ * LD;                       // valid only when DO_DK
  bf16_t* Qt = smem + N_RM * BQ * LD;               // valid only when DO_DK
  bf16_t* dOt = Qt + (DO_DK ? D * TS : 0);          // valid only when DO_DV
  float* lse_s = reinterpret_cast<float*>(smem + N_RM * BQ * LD + N_TR * D * TS);
  float* delta_s = lse_s + BQ;
  int* docq = reinterpret_cast<int*>(delta_s + BQ);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int kt = blockIdx.y, hk = blockIdx.x, b = blockIdx.z;   // (see head_of_slot: head in x, heavy tiles first)
  const int G = Nh / Nkv;
  const int k0 = kt * BNK;
  const int wk0 = k0 + 32 * wave;
  const int kvrow = wk0 + l31;
  const bool kvalid = kvrow < T;

  bf16x8_t kreg[KSTEPS], vreg[DO_DK ? KSTEPS : 1];
  {
    const size_t off = (((size_t)b * T + (kvalid ? kvrow : 0)) * Nkv + hk) * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
      if (kvalid) {
        a = *reinterpret_cast<const uint4*>(K + off + 16 * s);
        if (DO_DK) c = *reinterpret_cast<const uint4*>(V + off + 16 * s);
      }
      kreg[s] = as_bf16x8(a);
      if (DO_DK) vreg[s] = as_bf16x8(c);
    }
  }
  const int dkdoc = kvalid ? doc[(size_t)b * T + kvrow] : 0;
  int wminpos, wmax;
  wave_id_range(dkdoc, wminpos, wmax);
  const bool w_uniform = (wminpos == wmax) && !__any(dkdoc == 0);   // all 32 kv rows in one document

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = 2 * kt, t1 = min(2 * kt + 1, meta.nt - 1);
  const int bminpos = min(m_minpos[t0], m_minpos[t1]);
  const int bmax = max(m_max[t0], m_max[t1]);
  const int qhi64 = max(meta.kv_hi[(size_t)b * meta.nt + t0], meta.kv_hi[(size_t)b * meta.nt + t1]);
  const int qt_lo = k0 / BQ;                                  // first 64-row query tile (q >= kv), global index
  const int qt_end = min(qhi64 + 1, meta.nt);                 // exclusive
  // query tiles this launch owns: per segment, the global 64-tile range clipped to [qt_lo, qt_end)
  int seg_lo[2], seg_n[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int first = qv.off[s] / BQ, cnt = qv.tiles(s, BQ);
    seg_lo[s] = max(qt_lo, first);
    seg_n[s] = max(min(qt_end, first + cnt) - seg_lo[s], 0);
  }
  const int nqt = seg_n[0] + seg_n[1];
  const int n_it = nqt * G;                                   // flattened (head-in-group, q tile)
  auto tile_of = [&](int it, int& t64, int& lrow0, int& left) {   // global tile, local first row, rows left
    const int idx = it % nqt;
    const int s = idx >= seg_n[0] ? 1 : 0;
    t64 = seg_lo[s] + idx - (s ? seg_n[0] : 0);
    const int lt = t64 - qv.off[s] / BQ;
    lrow0 = qv.row0[s] + lt * BQ;
    left = min(qv.rows[s] - lt * BQ, T - t64 * BQ);
  };
  auto advance = [&](int it) {
    while (it < n_it) {
      int t64, l0, left;
      tile_of(it, t64, l0, left);
      if (tile_may_interact(m_minpos[t64], m_max[t64], bminpos, bmax)) break;
      ++it;
    }
    return it;
  };

  f32x16_t dkacc[DO_DK ? DBLK : 1], dvacc[DO_DV ? DBLK : 1];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (DO_DK) dkacc[i][r] = 0.f;
      if (DO_DV) dvacc[i][r] = 0.f;
    }

  const TLdsReader<BQ> trd(l31, hi);
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  TransposeStage<BQ, D, 256> qst, dost;   // one register image serves both LDS images of a tile
  float lse_st = 0.f, delta_st = 0.f;
  int doc_st = 0;
  const size_t qld = (size_t)Nh * D;
  auto issue = [&](int it) {
    int t64, lrow0, left;
    tile_of(it, t64, lrow0, left);
    const int h = hk * G + it / nqt;
    const size_t base = (((size_t)b * qv.rpb + lrow0) * Nh + h) * D;
    qst.load(Q + base, qld, left, tid);
    dost.load(dO + base, qld, left, tid);
    if (tid < BQ) {
      const bool ok = tid < left;
      const size_t si = ((size_t)b * Nh + h) * qv.rpb + lrow0 + (ok ? tid : 0);
      lse_st = ok ? LSE2[si] : INFINITY;
      delta_st = ok ? Delta[si] : 0.f;
      doc_st = ok ? doc[(size_t)b * T + t64 * BQ + tid] : 0;
    }
  };

  int it = advance(0);
  if (it < n_it) issue(it);
  while (it < n_it) {
    const int itn = advance(it + 1);
    __syncthreads();
    qst.store_rowmajor(Qs, tid);
    if (DO_DK) {
      qst.store(Qt, tid);
      dost.store_rowmajor(dOs, tid);
    }
    if (DO_DV) dost.store(dOt, tid);
    if (tid < BQ) {
      lse_s[tid] = lse_st;
      delta_s[tid] = delta_st;
      docq[tid] = doc_st;
    }
    __syncthreads();
    if (itn < n_it) issue(itn);

    int t64, lrow0_unused, left_unused;
    tile_of(it, t64, lrow0_unused, left_unused);
    const int qb = t64 * BQ;
    if (uniform(qb + BQ - 1 >= wk0 && tile_may_interact(m_minpos[t64], m_max[t64], wminpos, wmax))) {
      const bool q_uniform = w_uniform && m_min[t64] == m_max[t64] && m_max[t64] == wmax;
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const int qsb = qb + 32 * qs;
        if (uniform(qsb + 31 >= wk0)) {                       // else: every q of this half precedes the kv rows
          const bool need_mask = uniform(!(q_uniform && qsb >= wk0 + 31));
          // ---- S[q, kv] = Q K^T (; dP[q, kv] = dO V^T)   rows = q in registers, column = this lane's kv
          const bf16_t* qp = Qs + (32 * qs + l31) * LD + 8 * hi;
          f32x16_t sacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(qp)), kreg[0], zero16);
#pragma unroll
          for (int s = 1; s < KSTEPS; ++s)
            sacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(qp + 16 * s)), kreg[s], sacc);
          f32x16_t dpacc = zero16;
          if (DO_DK) {
            const bf16_t* dop = dOs + (32 * qs + l31) * LD + 8 * hi;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s)
              dpacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(dop + 16 * s)), vreg[s], dpacc);
          }
          float p[16];
          auto probs = [&](auto masked) {
            constexpr bool MASK = decltype(masked)::value;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int o = 32 * qs + 8 * r4 + 4 * hi;
              const float4 l4 = *reinterpret_cast<const float4*>(lse_s + o);
              const float le[4] = {l4.x, l4.y, l4.z, l4.w};
              int qd[4] = {0, 0, 0, 0};
              if (MASK) {
                const int4 q4 = *reinterpret_cast<const int4*>(docq + o);
                qd[0] = q4.x; qd[1] = q4.y; qd[2] = q4.z; qd[3] = q4.w;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float pv = fast_exp2(sacc[4 * r4 + e] * scale_log2 - le[e]);
                if (MASK) pv = ((kvrow <= qb + o + e) & (qd[e] == dkdoc) & (dkdoc > 0)) ? pv : 0.f;
                p[4 * r4 + e] = pv;
              }
            }
          };
          if (need_mask) probs(std::true_type{}); else probs(std::false_type{});
          if (DO_DV) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
              const u32x4_t t = {pack2bf(p[8 * sp + 0], p[8 * sp + 1]), pack2bf(p[8 * sp + 2], p[8 * sp + 3]),
                                 pack2bf(p[8 * sp + 4], p[8 * sp + 5]), pack2bf(p[8 * sp + 6], p[8 * sp + 7])};
              const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, t);
#pragma unroll
              for (int db = 0; db < DBLK; ++db)     // dV^T[d, kv] += dO^T[d, q] P[q, kv]
                dvacc[db] = mfma32(trd.operand(dOt, db, 8 * qs + 4 * sp), pb, dvacc[db]);
            }
          }
          if (DO_DK) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const float4 d4 = *reinterpret_cast<const float4*>(delta_s + 32 * qs + 8 * r4 + 4 * hi);
              const float de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) p[4 * r4 + e] *= dpacc[4 * r4 + e] - de[e];   // p becomes dS
            }
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
              const u32x4_t u = {pack2bf(p[8 * sp + 0], p[8 * sp + 1]), pack2bf(p[8 * sp + 2], p[8 * sp + 3]),
                                 pack2bf(p[8 * sp + 4], p[8 * sp + 5]), pack2bf(p[8 * sp + 6], p[8 * sp + 7])};
              const bf16x8_t dsb = __builtin_bit_cast(bf16x8_t, u);
#pragma unroll
              for (int db = 0; db < DBLK; ++db)     // dK^T[d, kv] += Q^T[d, q] dS[q, kv]
                dkacc[db] = mfma32(trd.operand(Qt, db, 8 * qs + 4 * sp), dsb, dkacc[db]);
            }
          }
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
        if (DO_DK) {
          o.x = pack2bf(dkacc[db][4 * r4 + 0] * scale, dkacc[db][4 * r4 + 1] * scale);
          o.y = pack2bf(dkacc[db][4 * r4 + 2] * scale, dkacc[db][4 * r4 + 3] * scale);
          *reinterpret_cast<uint2*>(dK + off + 32 * db + 8 * r4 + 4 * hi) = o;
        }
        if (DO_DV) {
          o.x = pack2bf(dvacc[db][4 * r4 + 0], dvacc[db][4 * r4 + 1]);
          o.y = pack2bf(dvacc[db][4 * r4 + 2], dvacc[db][4 * r4 + 3]);
          *reinterpret_cast<uint2*>(dV + off + 32 * db + 8 * r4 + 4 * hi) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dQ
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ Delta,
    bf16_t* __restrict__ dQ, const int* __restrict__ doc, AttnMeta meta, QView qv, int T, int Nh, int Nkv,
    float scale, float scale_log2) {
  constexpr int BM = 128, BN = 64;
  constexpr int KSTEPS = D / 16, DBLK = D / 32, LD = D + 8, TS = TLds<BN>::STRIDE;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BN * LD + D * TS + 2 * BN];
  bf16_t* Ks = smem;
  bf16_t* Vs = Ks + BN * LD;
  bf16_t* Kt = Vs + BN * LD;
  int* docs = reinterpret_cast<int*>(Kt + D * TS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int h = head_of_slot(blockIdx.x, Nh, Nkv), b = blockIdx.z;
  const int hk = h / (Nh / Nkv);
  int lq0, q0, qleft;
  qv.tile(gridDim.y - 1 - blockIdx.y, BM, lq0, q0, qleft);
  const int wq0 = q0 + 32 * wave;          // global position of the wave's first query row
  const int qrow = wq0 + l31;              // global position
  const int lrow = lq0 + 32 * wave + l31;  // row in the local Q / dO / dQ / LSE / delta buffers
  const bool qvalid = (32 * wave + l31 < qleft) && (qrow < T);

  bf16x8_t qreg[KSTEPS], doreg[KSTEPS];
  {
    const size_t off = (((size_t)b * qv.rpb + (qvalid ? lrow : 0)) * Nh + h) * D + 8 * hi;
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
  const float lse2 = qvalid ? LSE2[((size_t)b * Nh + h) * qv.rpb + lrow] : INFINITY;
  const float delta = qvalid ? Delta[((size_t)b * Nh + h) * qv.rpb + lrow] : 0.f;
  int wminpos, wmax;
  wave_id_range(dq, wminpos, wmax);
  const bool w_has_zero = __any(dq == 0);

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = q0 / kTile, t1 = min(t0 + 1, meta.nt - 1);
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

  const TLdsReader<BN> trd(l31, hi);
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
    kst.store_rowmajor(Ks, tid);
    kst.store(Kt, tid);
    vst.store(Vs, tid);
    if (tid < BN) docs[tid] = dstage;
    __syncthreads();
    if (jn <= j_hi) issue(jn);

    const int k0 = j * BN;
    if (uniform(k0 <= wq0 + 31 && tile_may_interact(wminpos, wmax, m_minpos[j], m_max[j]))) {
      const bool need_mask = uniform(!(m_min[j] == m_max[j] && m_max[j] == wminpos && wminpos == wmax &&
                                       !w_has_zero && (k0 + BN - 1 <= wq0)));
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        if (uniform(k0 + 32 * blk <= wq0 + 31)) {             // else: this 32-row KV block is above the diagonal
          const bf16_t* kp = Ks + (32 * blk + l31) * LD + 8 * hi;
          const bf16_t* vp = Vs + (32 * blk + l31) * LD + 8 * hi;
          f32x16_t sacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(kp)), qreg[0], zero16);
          f32x16_t dpacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(vp)), doreg[0], zero16);
#pragma unroll
          for (int s = 1; s < KSTEPS; ++s) {
            sacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(kp + 16 * s)), qreg[s], sacc);
            dpacc = mfma32(as_bf16x8(*reinterpret_cast<const uint4*>(vp + 16 * s)), doreg[s], dpacc);
          }
          float ds[16];
          auto dscore = [&](auto masked) {
            constexpr bool MASK = decltype(masked)::value;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              int dkk[4] = {0, 0, 0, 0};
              if (MASK) {
                const int4 dk = *reinterpret_cast<const int4*>(docs + 32 * blk + 8 * r4 + 4 * hi);
                dkk[0] = dk.x; dkk[1] = dk.y; dkk[2] = dk.z; dkk[3] = dk.w;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = 4 * r4 + e;
                float pv = fast_exp2(sacc[r] * scale_log2 - lse2);
                if (MASK) {
                  const int kv = k0 + 32 * blk + 8 * r4 + 4 * hi + e;
                  pv = ((kv <= qrow) & (dkk[e] == dq) & (dq > 0)) ? pv : 0.f;
                }
                ds[r] = pv * (dpacc[r] - delta);
              }
            }
          };
          if (need_mask) dscore(std::true_type{}); else dscore(std::false_type{});
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            const u32x4_t u = {pack2bf(ds[8 * sp + 0], ds[8 * sp + 1]), pack2bf(ds[8 * sp + 2], ds[8 * sp + 3]),
                               pack2bf(ds[8 * sp + 4], ds[8 * sp + 5]), pack2bf(ds[8 * sp + 6], ds[8 * sp + 7])};
            const bf16x8_t dsb = __builtin_bit_cast(bf16x8_t, u);
#pragma unroll
            for (int db = 0; db < DBLK; ++db)       // dQ^T[d, q] += K^T[d, kv] dS^T[kv, q]
              dqacc[db] = mfma32(trd.operand(Kt, db, 8 * blk + 4 * sp), dsb, dqacc[db]);
          }
        }
      }
    }
    j = jn;
  }

  if (qvalid) {
    bf16_t* op = dQ + (((size_t)b * qv.rpb + lrow) * Nh + h) * D;
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

static int attn_bwd_launch(const void* q, const void* k, const void* v, const void* o, const void* dout,
                           const float* lse2, float* delta, void* dq, void* dk, void* dv, const int* doc,
                           const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, QView qv,
                           void* stream) {
  if (B <= 0 || T <= 0 || Nh <= 0 || Nkv <= 0 || Nh % Nkv) return TN_EINVAL;
  if (D != 64 && D != 128) return TN_EINVAL;
  for (int s = 0; s < qv.nseg; ++s)
    if (qv.off[s] % 128 || qv.row0[s] % 128 || (s + 1 < qv.nseg && qv.rows[s] % 128)) return TN_EINVAL;
  const int nt = (T + kTile - 1) / kTile, n = B * nt;
  AttnMeta m = {meta, meta + n, meta + 2 * n, meta + 3 * n, meta + 4 * n, nt};
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  const size_t rows = (size_t)B * qv.rpb * Nh;
  dim3 gq(Nh, qv.tiles(0, 128) + qv.tiles(1, 128), B), gk(Nkv, (T + 127) / 128, B), block(256);
  const bf16_t *Q = (const bf16_t*)q, *K = (const bf16_t*)k, *V = (const bf16_t*)v, *dO = (const bf16_t*)dout;
  if (D == 128) {
    hipLaunchKernelGGL((attn_delta_kernel<128>), dim3((rows * 16 + 255) / 256), block, 0, st, (const bf16_t*)o, dO,
                       delta, B, qv.rpb, Nh);
    hipLaunchKernelGGL((attn_bwd_kv_kernel<128, 0>), gk, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dk,
                       (bf16_t*)dv, doc, m, qv, T, Nh, Nkv, scale, sl2);
    hipLaunchKernelGGL((attn_bwd_kv_kernel<128, 1>), gk, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dk,
                       (bf16_t*)dv, doc, m, qv, T, Nh, Nkv, scale, sl2);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<128>), gq, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dq, doc, m,
                       qv, T, Nh, Nkv, scale, sl2);
  } else {
    hipLaunchKernelGGL((attn_delta_kernel<64>), dim3((rows * 8 + 255) / 256), block, 0, st, (const bf16_t*)o, dO,
                       delta, B, qv.rpb, Nh);
    hipLaunchKernelGGL((attn_bwd_kv_kernel<64, 2>), gk, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dk,
                       (bf16_t*)dv, doc, m, qv, T, Nh, Nkv, scale, sl2);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), gq, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dq, doc, m,
                       qv, T, Nh, Nkv, scale, sl2);
  }
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// delta: float [B, Nh, T] scratch (also an output of this call).  dq/dk/dv are fully overwritten.
int tn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                int Nkv, int D, float scale, void* stream) {
  const QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T};
  return attn_bwd_launch(q, k, v, o, dout, lse2, delta, dq, dk, dv, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

// Sequence-sharded query side (context parallel), see tn_attn_fwd_seg.  q/o/dout/dq [B, rows_per_batch, Nh, D],
// lse2/delta [B, Nh, rows_per_batch]; k/v and the outputs dk/dv are GLOBAL [B, T, Nkv, D]: dk/dv receive this
// rank's partial sums over its own query rows (rows no local query can see are written as zeros) and are
// reduce-scattered over the CP group by the caller.
int tn_attn_bwd_seg(const void* q, const void* k, const void* v, const void* o, const void* dout,
                    const float* lse2, float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta,
                    int B, int T, int Nh, int Nkv, int D, float scale, int nseg, const int* segs, int rows_per_batch,
                    void* stream) {
  if (nseg < 1 || nseg > 2) return TN_EINVAL;
  const QView qv = {nseg, {segs[0], nseg > 1 ? segs[3] : 0}, {segs[1], nseg > 1 ? segs[4] : 0},
                    {segs[2], nseg > 1 ? segs[5] : 0}, rows_per_batch};
  return attn_bwd_launch(q, k, v, o, dout, lse2, delta, dq, dk, dv, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

}  // extern "C"
