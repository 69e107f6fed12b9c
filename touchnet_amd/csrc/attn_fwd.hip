// Packed (document-masked, causal) flash attention FORWARD for gfx950 — LDS-tiled, MFMA 32x32x16 bf16.
//
// Replaces the flex_attention / SDPA call the reference makes per decoder layer
// (SURVEY.md §2.3 K6/K6'; transformers/integrations/flex_attention.py:264-340 driven by the document-id
// `attention_mask` of touchnet/models/llama/processing_llama.py:38-40 and
// touchnet/models/touch_audio/processing_touch_audio.py:205-206; plain causal for the Qwen2-Audio path,
// touchnet/models/qwen2_audio/__init__.py:190-193,231-236 = all ids 1).
//
// Layout (chosen for the producing GEMMs, no transposes anywhere):
//   Q [B, T, Nh, D]   K, V [B, T, Nkv, D]   O [B, T, Nh, D]   bf16, D in {64, 128}
//   LSE2 [B, Nh, T] fp32 = log2-domain log-sum-exp of scale*log2(e)*S (+inf on fully masked rows)
//   doc [B, T] int32 document ids (0 = pad)
//
// Work decomposition: grid = (ceil(T/128), Nh, B); a 256-thread workgroup owns 128 query rows of one
// head, wave w owns rows 32w..32w+31.  KV is consumed in 64-row tiles; tiles whose document-id range
// cannot meet the query tile's range are never loaded (block-sparse over the packed batch), tiles fully
// inside one document strictly below the diagonal skip the per-element mask.
//
// Per KV tile and wave:  S^T = K Q^T (A = K from LDS, B = Q in registers) so that every lane owns ONE
// query column: softmax statistics are lane-local (one cross-half shuffle), P never leaves registers and
// feeds O^T += V^T P^T directly as the B operand (A = V^T, read with the gfx950 transpose read from the SAME kind of
// row-major panel image K uses: attn_common.h PTile).
#include <stdlib.h>

#include "attn_common.h"

namespace tn {

// ------------------------------------------------------------------------------------------------
// Metadata pre-pass: one thread per 64-position tile.
// ------------------------------------------------------------------------------------------------
__global__ void attn_meta_stats_kernel(const int* __restrict__ doc, int* tmin, int* tmax, int* tminpos, int B, int T,
                                       int nt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * nt) return;
  const int b = idx / nt, t = idx % nt;
  int mn = 0x7fffffff, mx = 0, mnp = 0x7fffffff;
  for (int i = 0; i < kTile; ++i) {
    const int p = t * kTile + i;
    const int d = p < T ? doc[(size_t)b * T + p] : 0;
    mn = min(mn, d);
    mx = max(mx, d);
    if (d > 0) mnp = min(mnp, d);
  }
  tmin[idx] = mn;
  tmax[idx] = mx;
  tminpos[idx] = mnp;
}

__global__ void attn_meta_range_kernel(const int* __restrict__ tmax, const int* __restrict__ tminpos, int* q_lo,
                                       int* kv_hi, int B, int nt) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * nt) return;
  const int b = idx / nt, t = idx % nt;
  const int* mx = tmax + (size_t)b * nt;
  const int* mp = tminpos + (size_t)b * nt;
  int lo = t + 1;
  for (int j = 0; j <= t; ++j)
    if (tile_may_interact(mp[t], mx[t], mp[j], mx[j])) {
      lo = j;
      break;
    }
  int hi = t - 1;
  for (int i = nt - 1; i >= t; --i)
    if (tile_may_interact(mp[i], mx[i], mp[t], mx[t])) {
      hi = i;
      break;
    }
  q_lo[idx] = lo;
  kv_hi[idx] = hi;
}

// Per 32 positions: the id statistics of one wave's query rows (attn_common.h qstat).
__global__ void attn_meta_qstat_kernel(const int* __restrict__ doc, int* __restrict__ qstat, int B, int T, int nq32) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * nq32) return;
  const int b = idx / nq32, g = idx % nq32;
  int mnp = 0x7fffffff, mx = 0, zero = 0;
  for (int i = 0; i < 32; ++i) {
    const int p = g * 32 + i;
    const int d = p < T ? doc[(size_t)b * T + p] : 0;
    mx = max(mx, d);
    if (d > 0) mnp = min(mnp, d);
    zero |= d == 0;
  }
  reinterpret_cast<int4*>(qstat)[idx] = make_int4(mnp, mx, zero, 0);
}

// Per 128-position causal query tile: the KV tiles it meets (attn_common.h klist) — one wave per tile, ballot compaction
// in tile order, i.e. exactly the list the forward / dQ workgroups build for themselves.
__global__ void attn_meta_klist_kernel(const int* __restrict__ tmin, const int* __restrict__ tmax,
                                       const int* __restrict__ tminpos, const int* __restrict__ q_lo,
                                       int* __restrict__ klist, int B, int nt, int nq128) {
  const int qt = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int* mn = tmin + (size_t)b * nt;
  const int* mx = tmax + (size_t)b * nt;
  const int* mp = tminpos + (size_t)b * nt;
  const int t0 = 2 * qt, t1 = min(2 * qt + 1, nt - 1);
  int bminpos = 0x7fffffff, bmax = 0, j_lo = nt;
  for (int t = t0; t <= t1; ++t) {
    bminpos = min(bminpos, mp[t]);
    bmax = max(bmax, mx[t]);
    j_lo = min(j_lo, q_lo[(size_t)b * nt + t]);
  }
  int4* out = reinterpret_cast<int4*>(klist + ((size_t)b * nq128 + qt) * (4 + 4 * kListPre));
  int count = 0;
  for (int base = j_lo; base <= t1; base += 64) {
    const int j = base + lane;
    const bool ok = j <= t1 && tile_may_interact(bminpos, bmax, mp[j <= t1 ? j : t1], mx[j <= t1 ? j : t1]);
    const unsigned long long bal = __ballot(ok);
    const int pos = count + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && pos < kListPre) out[1 + pos] = make_int4(j, mn[j], mx[j], mp[j]);
    count += __popcll(bal);
  }
  if (lane == 0) out[0] = make_int4(count, qt, 0, 0);
}

// Per 128-position KV tile: the 64-position query tiles it meets under the causal mask (attn_common.h qlist) — what the
// dK / dV workgroups derive for themselves in attn_bwd.hip (first tile = the diagonal, last = kv_hi of its two halves).
__global__ void attn_meta_qlist_kernel(const int* __restrict__ tmin, const int* __restrict__ tmax,
                                       const int* __restrict__ tminpos, const int* __restrict__ kv_hi,
                                       int* __restrict__ qlist, int B, int nt, int nq128) {
  const int kt = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int* mn = tmin + (size_t)b * nt;
  const int* mx = tmax + (size_t)b * nt;
  const int* mp = tminpos + (size_t)b * nt;
  const int t0 = 2 * kt, t1 = min(2 * kt + 1, nt - 1);
  const int bminpos = min(mp[t0], mp[t1]), bmax = max(mx[t0], mx[t1]);
  const int qt_end = min(max(kv_hi[(size_t)b * nt + t0], kv_hi[(size_t)b * nt + t1]) + 1, nt);      // exclusive
  int4* out = reinterpret_cast<int4*>(qlist + ((size_t)b * nq128 + kt) * (4 + 4 * kListPre));
  int count = 0;
  for (int base = t0; base < qt_end; base += 64) {
    const int t = base + lane;
    const int tc = t < qt_end ? t : t0;
    const bool ok = t < qt_end && tile_may_interact(mp[tc], mx[tc], bminpos, bmax);
    const unsigned long long bal = __ballot(ok);
    const int pos = count + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && pos < kListPre) out[1 + pos] = make_int4(t, mn[t], mx[t], mp[t]);
    count += __popcll(bal);
  }
  if (lane == 0) out[0] = make_int4(count, kt, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// ABL (ablation, timing experiments only — results are wrong for ABL != 0; reached through tn_attn_fwd_ablate):
//   1 no in-loop global loads / LDS stores   2 no softmax VALU   3 no P.V MFMAs   4 no QK^T MFMAs   5 no barrier
template <int D, int ABL = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, D == 64 ? 3 : 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                       float* __restrict__ LSE2, const int* __restrict__ doc,
                                                       AttnMeta meta, QView qv, int T, int Nh, int Nkv,
                                                       float scale_log2) {
  constexpr int BM = 32 * NW, BN = 64, NT = 64 * NW;   // NW waves x 32 query rows
  constexpr int KSTEPS = D / 16;   // MFMA k-steps over the head dim
  constexpr int DBLK = D / 32;     // 32-wide output blocks over the head dim
  using Tile = PTile<BN, D>;
  // two LDS buffers {K image | V image | doc ids}: tile j+1 is written while tile j is being consumed,
  // ONE barrier per KV tile (2 x 33.5 KB at D=128 -> two workgroups per CU)
  constexpr int BUF = 2 * Tile::SIZE + 2 * BN;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BUF];
  constexpr int CAP = 192;            // list chunk: 3 KB, keeps two workgroups per CU at D = 128 (2 x 79.4 KB)
  __shared__ __attribute__((aligned(16))) int4 tlist[CAP + 4];        // interacting KV tiles (attn_common.h)
  __shared__ int wcount[NW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int h = head_of_slot(blockIdx.x, Nh, Nkv), b = blockIdx.z;
  const int hk = h / (Nh / Nkv);
  int lq0, q0, qleft;                      // local first row / global first position / rows left in the segment
  qv.tile(gridDim.y - 1 - blockIdx.y, BM, lq0, q0, qleft);
  const int wq0 = q0 + 32 * wave;          // GLOBAL position of the wave's first query row
  const int qrow = wq0 + l31;              // global position: what the causal / document predicate compares
  const int lrow = lq0 + 32 * wave + l31;  // row in the local Q / O / LSE buffers
  const bool qvalid = (32 * wave + l31 < qleft) && (qrow < T);

  // ---- this lane's query row: MFMA B operand for every k-step, and its document id
  bf16x8_t qreg[KSTEPS];
  {
    const bf16_t* qp = Q + (((size_t)b * qv.rpb + (qvalid ? lrow : 0)) * Nh + h) * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qvalid) v = *reinterpret_cast<const uint4*>(qp + 16 * s);
      qreg[s] = as_bf16x8(v);
    }
  }
  const int dq = qvalid ? doc[(size_t)b * T + qrow] : 0;
  int wminpos, wmax;   // wave-level id range of the 32 query rows, in SGPRs
  wave_id_range(dq, wminpos, wmax);
  const bool w_has_zero = __any(dq == 0);

  // ---- block-level tile range from the metadata of the two 64-row halves of the query tile
  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = q0 / kTile, t1 = min(t0 + BM / kTile - 1, meta.nt - 1);
  int bminpos = 0x7fffffff, bmax = 0, j = meta.nt;
  for (int t = t0; t <= t1; ++t) {
    bminpos = min(bminpos, m_minpos[t]);
    bmax = max(bmax, m_max[t]);
    j = min(j, meta.q_lo[(size_t)b * meta.nt + t]);
  }
  int j_hi = t1;
  const int j_lo = j;
  const bool bidir = qv.bidir != 0;
  if (bidir)
    for (int t = t0; t <= t1; ++t) j_hi = max(j_hi, meta.kv_hi[(size_t)b * meta.nt + t]);
  const int qcap = bidir ? 0x7fffffff : qrow;          // `kv <= qcap`: the causal term of the predicate

  f32x16_t oacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const PRowReader<BN, D> krd(l31, hi);
  const PTrReader<BN, D> vrd(lane);
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  PStage<BN, D, NT> kst, vst;
  int dstage = 0;
  const size_t kvld = (size_t)Nkv * D;
  auto issue = [&](int jj) {
    const int k0 = jj * BN;
    const size_t base = (((size_t)b * T + k0) * Nkv + hk) * D;
    kst.load(K + base, kvld, T - k0, tid);
    vst.load(V + base, kvld, T - k0, tid);
    if (tid < BN) dstage = (k0 + tid < T) ? doc[(size_t)b * T + k0 + tid] : 0;
  };
  auto stage_store = [&](int buf) {
    bf16_t* base = smem + buf * BUF;
    kst.store(base, tid);
    vst.store(base + Tile::SIZE, tid);
    if (tid < BN) reinterpret_cast<int*>(base + 2 * Tile::SIZE)[tid] = dstage;
  };
  // The tiles of [j_lo, j_hi] are walked in chunks of CAP through the LDS list.
  int cur = 0;
  for (int c_lo = j_lo; c_lo <= j_hi; c_lo += CAP) {
    const int n = build_kv_list<NT>(tlist, wcount, c_lo, min(c_lo + CAP - 1, j_hi), j_hi + 1, bminpos, bmax, m_min,
                                    m_max, m_minpos, tid, qv.kv_tpc, qv.kv_mask);
    if (n == 0) continue;
    issue(list_entry(tlist, 0).x);
    stage_store(cur);
    int4 e_cur = list_entry(tlist, 0), e_nxt = list_entry(tlist, 1);
    if (e_nxt.x <= j_hi) issue(e_nxt.x);    // tile i+1 flies under the compute of tile i
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const int4 e_nn = tlist[i + 2];       // (vector read now, scalarised at the hand-over)
      const bf16_t* Ks = smem + cur * BUF;
      const bf16_t* Vs = Ks + Tile::SIZE;
      const int* docs = reinterpret_cast<const int*>(Vs + Tile::SIZE);
      const int j = e_cur.x, kmin = e_cur.y, kmax = e_cur.z, kminpos = e_cur.w;
      const int k0 = j * BN;
    if (uniform((bidir || k0 <= wq0 + 31) && tile_may_interact(wminpos, wmax, kminpos, kmax))) {
      const bool need_mask = uniform(
          !(kmin == kmax && kmax == wminpos && wminpos == wmax && !w_has_zero && (bidir || k0 + BN - 1 <= wq0)));
      // ---- S^T[kv, q] = K[kv, :] . Q[q, :]
      f32x16_t sacc[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        sacc[blk] = mfma32(krd.operand(Ks, 32 * blk, 0), qreg[0], zero16);
#pragma unroll
        for (int s = 1; s < (ABL == 4 ? 1 : KSTEPS); ++s)
          sacc[blk] = mfma32(krd.operand(Ks, 32 * blk, s), qreg[s], sacc[blk]);
      }
      // ---- mask, online softmax (lane-local: this lane's query column).  Scores stay RAW in the accumulator;
      // the softmax scale rides in the exponent's fma: p = exp2(s * c - m), m tracked in the scaled domain.
      float mx = -INFINITY;
      if (need_mask) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int4 dk = *reinterpret_cast<const int4*>(docs + 32 * blk + 8 * r4 + 4 * hi);
            const int dkk[4] = {dk.x, dk.y, dk.z, dk.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int kv = k0 + 32 * blk + 8 * r4 + 4 * hi + e;
              const bool ok = (kv <= qcap) & (dkk[e] == dq) & (dq > 0);
              sacc[blk][4 * r4 + e] = ok ? sacc[blk][4 * r4 + e] : -INFINITY;
            }
          }
        }
      }
      {   // four independent v_max3 chains (fmaxf() canonicalises every MFMA output first: 2x the VALU work)
        float mxs[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int blk = c >> 1, o = 8 * (c & 1);
          mxs[c] = max3(sacc[blk][o + 0], sacc[blk][o + 1], sacc[blk][o + 2]);
          mxs[c] = max3(mxs[c], sacc[blk][o + 3], sacc[blk][o + 4]);
          mxs[c] = max3(mxs[c], sacc[blk][o + 5], sacc[blk][o + 6]);
        }
        mx = max3(mxs[0], mxs[1], sacc[0][7]);
        mx = max3(mx, mxs[2], sacc[0][15]);
        mx = max3(mx, mxs[3], sacc[1][7]);
        mx = max3(mx, sacc[1][15], sacc[1][15]);
      }
      mx = half_swap_max(mx) * scale_log2;
      if (ABL == 2) mx = m_run;
      // Deferred rescale (threshold 8 in the log2 domain): while no row's running max grows by more than 2^8 the
      // old reference max stays, P <= 256 is exact enough in bf16 and the 16*DBLK-register O rescale is skipped.
      float alpha = 1.f;
      if (uniform(!__all(mx - m_run <= 8.f))) {
        const float m_new = fmaxf(m_run, mx);
        alpha = fast_exp2(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < DBLK; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
      }
      const float neg_m = -m_run;
      float psum = 0.f;
      bf16x8_t pb[2][2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float p[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            p[e] = ABL == 2 ? sacc[blk][8 * sp + e] : fast_exp2(fmaf(sacc[blk][8 * sp + e], scale_log2, neg_m));
            if (ABL != 2) psum += p[e];
          }
          u32x4_t t = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
          pb[blk][sp] = __builtin_bit_cast(bf16x8_t, t);
        }
      }
      l_run += psum;
      // ---- O^T[d, q] += V^T[d, kv] P^T[kv, q]
#pragma unroll
      for (int db = 0; db < DBLK; ++db) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            if (ABL == 3 && (db | blk | sp)) {
              asm volatile("" ::"v"(pb[blk][sp]));      // keep P live so its producers are not dead code
              continue;
            }
            oacc[db] = mfma32(vrd.operand(Vs, db, 32 * blk + 16 * sp), pb[blk][sp], oacc[db]);
          }
        }
      }
    }
      // ---- hand-over: tile i+1 (already in registers) -> the other buffer, then prefetch tile i+2.
      // WAR-safe: the other buffer was last read in the previous iteration, which every wave left through the
      // barrier below; RAW-safe: it is read only after this iteration's barrier.
      e_cur = e_nxt;
      e_nxt = scalarize(e_nn);
      if (i + 1 < n) {
        if (ABL != 1) stage_store(cur ^ 1);
        if (e_nxt.x <= j_hi && i + 2 < n && ABL != 1) issue(e_nxt.x);
      }
      if (ABL != 5) __syncthreads();
      cur ^= 1;
    }
  }

  // ---- epilogue: normalise, store O (4 consecutive head-dim elements = 8 bytes per store) and LSE2
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qvalid) {
    bf16_t* op = O + (((size_t)b * qv.rpb + lrow) * Nh + h) * D;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        uint2 o;
        o.x = pack2bf(oacc[db][4 * r4 + 0] * inv, oacc[db][4 * r4 + 1] * inv);
        o.y = pack2bf(oacc[db][4 * r4 + 2] * inv, oacc[db][4 * r4 + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * db + 8 * r4 + 4 * hi) = o;
      }
    }
    if (hi == 0) LSE2[((size_t)b * Nh + h) * qv.rpb + lrow] = l_tot > 0.f ? m_run + log2f(l_tot) : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------
// Merge of two partial attention results over DISJOINT key sets (context parallel: own chunks / received chunks):
//   lse = log2(2^lse_a + 2^lse_b),  O = (2^lse_a O_a + 2^lse_b O_b) / 2^lse      (LSE2 convention of the forward kernels:
// log2 domain, +inf = the row saw no key in that part).  HBM-bound: 3 x rows x Nh x D x 2 B + the statistics.
// The reference's ring attention merges its per-step (out, lse) the same way
// (torch/distributed/tensor/experimental/_context_parallel/_attention.py:182-183 via touchnet/utils/distributed.py:292-315).
// ------------------------------------------------------------------------------------------------
__global__ void attn_merge_kernel(const bf16_t* __restrict__ oa, const float* __restrict__ la,
                                  const bf16_t* __restrict__ ob, const float* __restrict__ lb, bf16_t* __restrict__ o,
                                  float* __restrict__ l, int B, int R, int Nh, int D) {
  const int dv = D / 8;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * R * Nh * dv;
  if (idx >= total) return;
  const int d8 = (int)(idx % dv);
  const int h = (int)((idx / dv) % Nh);
  const size_t br = idx / ((size_t)dv * Nh);                 // b * R + r
  const int b = (int)(br / R), r = (int)(br % R);
  const size_t li = ((size_t)b * Nh + h) * R + r;
  const float a = la[li], c = lb[li];
  const bool ea = a == INFINITY, ec = c == INFINITY;          // empty parts
  const float m = ea ? c : (ec ? a : fmaxf(a, c));
  const float wa = ea ? 0.f : fast_exp2(a - m), wc = ec ? 0.f : fast_exp2(c - m);
  const float tot = wa + wc;
  const float inv = tot > 0.f ? 1.f / tot : 0.f;
  Vec16<bf16_t> va, vc, vo;
  va.load(oa + idx * 8);
  vc.load(ob + idx * 8);
  float fa[8], fc[8], fo[8];
  va.unpack(fa);
  vc.unpack(fc);
#pragma unroll
  for (int e = 0; e < 8; ++e) fo[e] = (wa * fa[e] + wc * fc[e]) * inv;
  vo.pack(fo);
  vo.store(o + idx * 8);
  if (d8 == 0) l[li] = tot > 0.f ? m + log2f(tot) : INFINITY;
}

}  // namespace tn

using namespace tn;

int tn_attn_fwd_pp_launch(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                          AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, int D, float sl2, hipStream_t st);
// attn_fwd_stream.hip: precomputed tile lists, K / V / Q by LDS-DMA, whole-row stores
int tn_attn_fwd_stream_launch(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                              AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, int D, float sl2, hipStream_t st);

// Schedule selection: 2 = attn_fwd_stream.hip (default: 14-20 % faster than schedule 0 on every shape measured,
// profiles/r06a_attn_fwd_stream_vs_base_same_box.log); 1 = ping-pong (attn_fwd_pp.hip: 256-row workgroups, the fastest on
// the long-sequence recipes — T >= 32768, D = 128: config D's 20-minute recordings, 8.81 vs 9.00 ms at T = 32768 plain
// causal); 0 = this file's kernel (register-staged K / V tiles, everything derived inside the workgroup: kept as the
// reference the other two are compared against, and for the ablation entry point).  TN_ATTN_FWD_SCHEDULE = 0 / 1 / 2
// forces one; unset = by shape.
static int g_fwd_schedule_override = -2;     // tn_attn_set_fwd_schedule (development entry point)
static int fwd_schedule(int T, int D) {
  static int mode = [] {
    const char* e = getenv("TN_ATTN_FWD_SCHEDULE");
    return e ? atoi(e) : -1;
  }();
  const int m = g_fwd_schedule_override >= -1 ? g_fwd_schedule_override : mode;
  return m >= 0 ? m : (T >= 32768 && D == 128 ? 1 : 2);
}

template <int ABL, int NW = 4>
static int attn_fwd_launch_abl(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                               AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, float sl2, hipStream_t st) {
  dim3 grid(Nh, qv.tiles(0, 32 * NW) + qv.tiles(1, 32 * NW), B), block(64 * NW);
  hipLaunchKernelGGL((attn_fwd_kernel<128, ABL, NW>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (bf16_t*)o, lse2, doc, m, qv, T, Nh, Nkv, sl2);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

extern "C" {

// meta buffers: 5 int32 arrays of B*nt each, nt = ceil(T/64):  [tmin | tmax | tminpos | q_lo | kv_hi], then
// [qstat | klist] (attn_common.h AttnMeta)
int tn_attn_meta_ints(int B, int T) { return attn_meta_ints(B, T); }

int tn_attn_build_meta(const int* doc, int* meta, int B, int T, void* stream) {
  if (B <= 0 || T <= 0) return TN_EINVAL;
  const int nt = (T + kTile - 1) / kTile, n = B * nt;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_meta_stats_kernel, dim3((n + 127) / 128), dim3(128), 0, st, doc, meta, meta + n,
                     meta + 2 * n, B, T, nt);
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_meta_range_kernel, dim3((n + 127) / 128), dim3(128), 0, st, meta + n, meta + 2 * n,
                     meta + 3 * n, meta + 4 * n, B, nt);
  TN_LAUNCH_CHECK();
  const AttnMeta m = make_attn_meta(meta, B, T);
  hipLaunchKernelGGL(attn_meta_qstat_kernel, dim3((B * m.nq32 + 127) / 128), dim3(128), 0, st, doc,
                     const_cast<int*>(m.qstat), B, T, m.nq32);
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_meta_klist_kernel, dim3(m.nq128, B), dim3(64), 0, st, m.tmin, m.tmax, m.tminpos, m.q_lo,
                     const_cast<int*>(m.klist), B, nt, m.nq128);
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_meta_qlist_kernel, dim3(m.nq128, B), dim3(64), 0, st, m.tmin, m.tmax, m.tminpos, m.kv_hi,
                     const_cast<int*>(m.qlist), B, nt, m.nq128);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

static int attn_fwd_launch(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                           const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, QView qv,
                           void* stream) {
  if (B <= 0 || T <= 0 || Nh <= 0 || Nkv <= 0 || Nh % Nkv) return TN_EINVAL;
  for (int s = 0; s < qv.nseg; ++s)
    if (qv.off[s] % 128 || qv.row0[s] % 128 || (s + 1 < qv.nseg && qv.rows[s] % 128)) return TN_EINVAL;
  const int nt = (T + kTile - 1) / kTile;
  const AttnMeta m = make_attn_meta(meta, B, T);
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  int sched = fwd_schedule(T, D);
  if (sched == 1 && (qv.bidir || nt > 1024)) sched = 2;     // (the ping-pong kernel: causal, its LDS tile list)
  if (sched == 1 && (D == 64 || D == 128))
    return tn_attn_fwd_pp_launch(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, D, sl2, st);
  if (sched == 2 && (D == 64 || D == 128))
    return tn_attn_fwd_stream_launch(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, D, sl2, st);
  dim3 grid(Nh, qv.tiles(0, 128) + qv.tiles(1, 128), B), block(256);
  if (D == 128)
    hipLaunchKernelGGL((attn_fwd_kernel<128>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)o, lse2, doc, m, qv, T, Nh, Nkv, sl2);
  else if (D == 64)
    hipLaunchKernelGGL((attn_fwd_kernel<64>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)o, lse2, doc, m, qv, T, Nh, Nkv, sl2);
  else
    return TN_EINVAL;
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc, const int* meta,
                int B, int T, int Nh, int Nkv, int D, float scale, void* stream) {
  const QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T, 0, ~0ull};
  return attn_fwd_launch(q, k, v, o, lse2, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

// Bidirectional inside a document (no causal term): the Whisper speech encoder of Kimi-Audio.
int tn_attn_fwd_bidir(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                      const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, void* stream) {
  QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T, 0, ~0ull};
  qv.bidir = 1;
  return attn_fwd_launch(q, k, v, o, lse2, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

// Development entry point (NOT part of the C ABI): force a forward schedule for A/B runs inside one process.
// mode -2 = back to TN_ATTN_FWD_SCHEDULE / the shape rule, -1 = the shape rule, 0 / 1 / 2 = a schedule.
int tn_attn_set_fwd_schedule(int mode) {
  g_fwd_schedule_override = mode;
  return TN_OK;
}

// Timing experiments only (D = 128): the forward with one piece removed, see ABL above.  Output is garbage.
// Development entry point: exported, but NOT declared in include/touchnet_amd.h (not part of the C ABI).
int tn_attn_fwd_ablate(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                       const int* meta, int B, int T, int Nh, int Nkv, float scale, int ablation, void* stream) {
  const AttnMeta m = make_attn_meta(meta, B, T);
  const QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T, 0, ~0ull};
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  switch (ablation) {
    case 0: return attn_fwd_launch_abl<0>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);
    case 1: return attn_fwd_launch_abl<1>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);
    case 2: return attn_fwd_launch_abl<2>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);
    case 3: return attn_fwd_launch_abl<3>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);
    case 4: return attn_fwd_launch_abl<4>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);
    case 5: return attn_fwd_launch_abl<5>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);
    case 6: return attn_fwd_launch_abl<0, 8>(q, k, v, o, lse2, doc, m, qv, B, T, Nh, Nkv, sl2, st);   // 8 waves
    default: return TN_EINVAL;
  }
}

// Sequence-sharded query side (context parallel): q / o are [B, rows_per_batch, Nh, D], lse2 [B, Nh, rows_per_batch];
// segs = host int[6] {row0_a, rows_a, off_a, row0_b, rows_b, off_b}; k / v / doc / meta stay global ([B, T, ...]).
int tn_attn_fwd_seg(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                    const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, int nseg, const int* segs,
                    int rows_per_batch, void* stream) {
  if (nseg < 1 || nseg > 2) return TN_EINVAL;
  const QView qv = {nseg, {segs[0], nseg > 1 ? segs[3] : 0}, {segs[1], nseg > 1 ? segs[4] : 0},
                    {segs[2], nseg > 1 ? segs[5] : 0}, rows_per_batch, 0, ~0ull};
  return attn_fwd_launch(q, k, v, o, lse2, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

// The same, restricted on the KEY side to the sequence chunks whose bit is set in `chunk_mask` (chunk c = positions
// [c * chunk_len, (c + 1) * chunk_len); chunk_len a multiple of 64, at most 64 chunks).  Rows that see no key in those
// chunks get O = 0 and LSE2 = +inf, which tn_attn_merge treats as "no contribution".
int tn_attn_fwd_seg_chunks(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                           const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, int nseg, const int* segs,
                           int rows_per_batch, int chunk_len, unsigned long long chunk_mask, void* stream) {
  if (nseg < 1 || nseg > 2 || chunk_len <= 0 || chunk_len % kTile || (T + chunk_len - 1) / chunk_len > 64) return TN_EINVAL;
  const QView qv = {nseg, {segs[0], nseg > 1 ? segs[3] : 0}, {segs[1], nseg > 1 ? segs[4] : 0},
                    {segs[2], nseg > 1 ? segs[5] : 0}, rows_per_batch, chunk_len / kTile, chunk_mask};
  return attn_fwd_launch(q, k, v, o, lse2, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

// O / LSE2 of two partial attentions over disjoint key sets -> merged (o may alias o_a, lse2 may alias lse2_a).
// o_* [B, rows, Nh, D] bf16, lse2_* [B, Nh, rows] fp32 (log2 domain, +inf = no key seen).
int tn_attn_merge(const void* o_a, const float* lse2_a, const void* o_b, const float* lse2_b, void* o, float* lse2,
                  int B, int rows, int Nh, int D, void* stream) {
  if (B <= 0 || rows <= 0 || Nh <= 0 || D <= 0 || D % 8) return TN_EINVAL;
  const size_t total = (size_t)B * rows * Nh * (D / 8);
  hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)o_a, lse2_a, (const bf16_t*)o_b, lse2_b, (bf16_t*)o, lse2, B, rows, Nh, D);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

}  // extern "C"
