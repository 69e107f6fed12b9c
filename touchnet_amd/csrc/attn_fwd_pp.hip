// Packed (document-masked, causal) flash attention FORWARD — "ping-pong" variant for gfx950.
//
// Same contract, layouts and MFMA conventions as attn_fwd.hip; what changes is the SCHEDULE.  Ablation of the
// baseline kernel (scripts/attn_ablate.py) showed that the phases of a wave (QK^T MFMAs | softmax VALU | P.V
// MFMAs | staging | barrier) add up to the total: with 2 waves per SIMD nothing overlaps, and an 8-wave block in
// lockstep is even slower.  Here a 512-thread workgroup owns 256 query rows as two groups of 4 waves
// (one wave of each group per SIMD) that run ONE PHASE APART:
//
//     phase p:   group A   [ P.V(t-1) , QK^T(t) ]  MFMA-heavy      group B   softmax(t-1) + stage K   VALU/LDS-heavy
//     phase p+1: group A   softmax(t) + stage V                    group B   [ P.V(t-1) , QK^T(t) ]
//
// so every SIMD always has one wave feeding the matrix pipe while its partner does exp/cvt/LDS work — the
// matrix-beside-memory pairing MI355X_MICROARCH.md ("Two waves per SIMD") asks for.  One workgroup barrier per
// phase; both groups execute the same code, group B simply passes one extra barrier before entering the loop.  The staging unit is skewed to match: {K(t+1), V(t)} becomes visible together (group B writes K(t+1)
// during phase 2t, group A writes V(t) during phase 2t+1; both are first read in phase 2t+2), two LDS slots each.
#include <type_traits>

#include "attn_common.h"

namespace tn {

#ifdef TN_PP_TRACE   // kernel-development instrumentation (scripts/build_variant.sh ... -DTN_PP_TRACE), never in the product build
__device__ unsigned long long g_pp_trace[2][192][8];
__device__ unsigned long long g_pp_blocks[8192][2];   // per workgroup: s_memrealtime at entry / exit
#define PP_STAMP(slot)                                                                          \
  do {                                                                                          \
    if (trace_on && t < 192) g_pp_trace[grp][t][slot] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define PP_STAMP(slot) \
  do {                 \
  } while (0)
#endif

// kernel-development ablations (variants only): -DTN_PP_NOMFMA keeps the operand reads but drops the MFMAs,
// -DTN_PP_NOLDS keeps the MFMAs but feeds them registers instead of LDS operands.  Output is garbage.
#if defined(TN_PP_NOMFMA)
#define PP_MFMA(a, b, c) (keep_alive(a), (c))
#else
#define PP_MFMA(a, b, c) mfma32(as_bf16x8(a), b, c)
#endif
#if defined(TN_PP_NOLDS)
#define PP_LDS16(p, dflt) (dflt)
#else
#define PP_LDS16(p, dflt) (*reinterpret_cast<const uint4*>(p))
#endif
__device__ __forceinline__ void keep_alive(uint4 v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }

// LDS image of V^T for the P.V operand: element (d, kv) at  d * (R + 8) + 4 * pos(kv >> 2) + (kv & 3)  with
// pos(g) = 4 * (g >> 2) + bitswap2(g & 3): the two 4-kv groups {g, g + 2} that one lane-half contracts in one MFMA
// (rows 4*hi + {0..3} and + 8 of the QK^T C-layout) sit side by side, so an operand is ONE ds_read_b128 (256 B/clk,
// no v_mov re-packing) instead of two ds_read2_b64 halves (128 B/clk).  Row stride (R+8)*2 B = 4 dwords mod 64:
// the b128 lane groups of MI355X_MICROARCH.md's LDS table hit 16 distinct 4-bank sets; the 8-byte transposing
// stores of 16 consecutive lanes (16 kv groups of one d) fill one contiguous 128-B row: conflict-free both ways.
__device__ __forceinline__ constexpr int vt_pos(int g) { return 4 * (g >> 2) + (((g & 1) << 1) | ((g >> 1) & 1)); }

template <int N>
using IC = std::integral_constant<int, N>;

constexpr int kMaxList = 1024;              // KV tiles per batch row the tile list holds (T <= 65536)

template <int D>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                             const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                             float* __restrict__ LSE2, const int* __restrict__ doc,
                                                             AttnMeta meta, QView qv, int T, int Nh, int Nkv,
                                                             float scale_log2) {
  constexpr int BM = 256, BN = 64;
  constexpr int KSTEPS = D / 16, DBLK = D / 32, KLD = D + 8, VLD = BN + 8, CPR = D / 8;
  constexpr int KSLOT = BN * KLD + 2 * BN;            // K row-major + the tile's doc ids (int32 x 64)
  constexpr int VSLOT = D * VLD;
  __shared__ __attribute__((aligned(16))) bf16_t smem[3 * KSLOT + 2 * VSLOT];
  __shared__ __attribute__((aligned(16))) int4 tlist[kMaxList + 4];   // {tile, min id, max id, min positive id}
  __shared__ int wcount[8];
  // D = 64 has 128 staging units for a group's 256 threads: the idle half runs the same (branch-free) pieces with
  // zero-filled loads and stores into this dump area
  constexpr bool ALL_STAGE = (BN / 4) * CPR == 256;
  __shared__ __attribute__((aligned(16))) bf16_t dump[ALL_STAGE ? 8 : 4 * KLD + 256 * 8 + 8];
  bf16_t* const Kbuf = smem;
  bf16_t* const Vbuf = smem + 3 * KSLOT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef TN_PP_TRACE
  const int lin_block = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (tid == 0 && lin_block < 8192) {
    g_pp_blocks[lin_block][0] = __builtin_amdgcn_s_memrealtime();
    g_pp_trace[0][191][0] = __builtin_amdgcn_s_memtime();       // (clock-ratio probe, any block)
    g_pp_trace[0][191][1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);   // 0 = group A, 1 = group B (one phase behind); SGPR
  const int gtid = tid & 255;               // thread index inside the group (staging work split)
  const int l31 = lane & 31, hi = lane >> 5;
  const int h = head_of_slot(blockIdx.x, Nh, Nkv), b = blockIdx.z;
  const int hk = h / (Nh / Nkv);
  int lq0, q0, qleft;
  qv.tile(gridDim.y - 1 - blockIdx.y, BM, lq0, q0, qleft);
  const int wq0 = q0 + 32 * wave;
  const int qrow = wq0 + l31;
  const int lrow = lq0 + 32 * wave + l31;
  const bool qvalid = (32 * wave + l31 < qleft) && (qrow < T);

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
  int wminpos, wmax;
  wave_id_range(dq, wminpos, wmax);
  const bool w_has_zero = __any(dq == 0);

  // ---- block-level list of KV tiles, built ONCE into LDS.  Walking the metadata arrays inside the loop costs
  // two dependent scalar global loads per tile and phase (~1.2k cycles each way in the s_memtime trace), which
  // the lock-step schedule cannot hide.
  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = q0 / kTile, t1 = min(t0 + BM / kTile - 1, meta.nt - 1);
  int bminpos = 0x7fffffff, bmax = 0, jlo = meta.nt;
  for (int t = t0; t <= t1; ++t) {
    bminpos = min(bminpos, m_minpos[t]);
    bmax = max(bmax, m_max[t]);
    jlo = min(jlo, meta.q_lo[(size_t)b * meta.nt + t]);
  }
  const int j_hi = t1;
  int n = 0;
  for (int base = jlo; base <= j_hi; base += 512) {
    const int j = base + tid;
    int mn = 0, mx = 0, mp = 0;
    bool ok = false;
    if (j <= j_hi) {
      mn = m_min[j];
      mx = m_max[j];
      mp = m_minpos[j];
      ok = tile_may_interact(bminpos, bmax, mp, mx) && qv.kv_tile_on(j);    // (context parallel: own / received chunks only)
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int before = n, total = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int c = wcount[w];
      before += w < wave ? c : 0;
      total += c;
    }
    if (ok) tlist[before + __popcll(bal & ((1ull << lane) - 1ull))] = make_int4(j, mn, mx, mp);
    n += __builtin_amdgcn_readfirstlane(total);
    __syncthreads();
  }
  if (tid < 4) tlist[n + tid] = make_int4(j_hi + 1, 0, 0, 0);   // sentinels: "no tile"
  __syncthreads();

  f32x16_t oacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t kvld = (size_t)Nkv * D;

  // per-wave state that crosses phases
  f32x16_t sacc[2];          // raw scores of the tile between its QK^T and its softmax
  bf16x8_t pb[2][2];         // bf16 probabilities of the tile between its softmax and its P.V
  bool act_s = false;        // the tile held in sacc is active for this wave
  bool act_p = false;        // the tile held in pb is active for this wave

  // ---- staging: group A stages V tiles (transposed image), group B stages K tiles (row-major image + doc ids).
  // ONE load path and ONE register image for both (source pointer and per-thread offsets are the only
  // difference), so the loaded registers have a single definition per loop iteration: a per-group load path made
  // the compiler copy the registers right after the loads were issued (phi copies = a wait for HBM every phase).
  // A thread owns a unit of 4 tile rows x 8 columns:  group B: unit = (rows 4*(gtid/CPR).., chunk gtid%CPR)
  // (16-byte row-major stores of a wave cover whole rows);  group A: unit = (kv group gtid%16, chunk gtid/16)
  // (8-byte transposing stores of 16 consecutive lanes fill one V^T row).
  const int u_g = grp == 0 ? (gtid & 15) : gtid / CPR;        // 4-row group of the tile
  const int u_c = grp == 0 ? (gtid >> 4) : gtid % CPR;        // 8-column chunk
  const bool u_ok = u_c < CPR && u_g < BN / 4;
  uint32_t soff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) soff[k] = u_ok ? (uint32_t)(((size_t)(4 * u_g + k) * kvld + u_c * 8) * 2) : 0xffffffffu;
  uint4 stg[4];
  int dstage = 0;
  const size_t head_off = ((size_t)b * T * Nkv + hk) * D;
  const bf16_t* const stage_src = (grp == 0 ? V : K) + head_off;
  // The staging of a tile is cut in pieces that ride in the gaps between the MFMAs of the wave's own MFMA phase
  // (the s_memtime trace showed the softmax alone filling the VALU phase: ~150 VALU at 4-12 cycles each):
  //   store pieces 0..7 (P.V loop):  A: 2 v_perm + one 8-byte transposing store each;  B: 4 row stores + doc ids
  //   issue pieces 0..4 (QK^T loop): the 4 tile loads + the doc-id load of the NEXT tile into the same registers
  __amdgpu_buffer_rsrc_t st_rs, st_drs;
  auto issue_setup = [&](const bf16_t* src, int jj) __attribute__((always_inline)) {   // jj > j_hi: no tile -> zero rows, no memory traffic
    const int k0 = jj * BN;
    const bool live = jj <= j_hi;
    const int left = live ? min(T - k0, BN) : 0;
    st_rs = tile_rsrc(src + (size_t)(live ? k0 : 0) * kvld, kvld, left, D);
    // doc ids of the tile through a buffer descriptor: rows past T (or no tile) read 0 = the pad id
    st_drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(doc + (size_t)b * T + (live ? k0 : 0)), 0,
                                               __builtin_amdgcn_readfirstlane(4 * left), 0x00020000);
  };
  auto issue_piece = [&](int i) __attribute__((always_inline)) {
    if (i < 4) stg[i] = buf_load16(st_rs, soff[i]);
    if (i == 4) dstage = __builtin_amdgcn_raw_buffer_load_b32(st_drs, 4 * (gtid & (BN - 1)), 0, 0);
  };
  // (group as a compile-time constant and no per-lane condition: a scalar branch per piece broke the MFMA loop
  // into basic blocks and cost ~100 cycles a piece)
  auto store_piece = [&](auto G, int i, bf16_t* kbase, bf16_t* vbase, int* dbase) __attribute__((always_inline)) {
    if constexpr (decltype(G)::value == 0) {
      const uint32_t w0[4] = {stg[0].x, stg[0].y, stg[0].z, stg[0].w}, w1[4] = {stg[1].x, stg[1].y, stg[1].z, stg[1].w};
      const uint32_t w2[4] = {stg[2].x, stg[2].y, stg[2].z, stg[2].w}, w3[4] = {stg[3].x, stg[3].y, stg[3].z, stg[3].w};
      const int wi = i >> 1;
      const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;
      uint2 o;
      o.x = __builtin_amdgcn_perm(w1[wi], w0[wi], sel);
      o.y = __builtin_amdgcn_perm(w3[wi], w2[wi], sel);
      *reinterpret_cast<uint2*>(vbase + i * VLD) = o;
    } else {
      if (i < 4) *reinterpret_cast<uint4*>(kbase + i * KLD) = stg[i];
      if (i == 4) *dbase = dstage;
    }
  };
  // this thread's spots in a slot (idle stagers: the dump area)
  int* const dump_i = reinterpret_cast<int*>(dump);
  auto k_spot = [&](int slot) __attribute__((always_inline)) {
    return (ALL_STAGE || u_ok) ? Kbuf + slot * KSLOT + (4 * u_g) * KLD + u_c * 8 : dump + gtid * 8;
  };
  auto v_spot = [&](int slot) __attribute__((always_inline)) {
    return (ALL_STAGE || u_ok) ? Vbuf + slot * VSLOT + (u_c * 8) * VLD + 4 * vt_pos(u_g) : dump + gtid * 4;
  };
  auto d_spot = [&](int slot) __attribute__((always_inline)) {
    return gtid < BN ? reinterpret_cast<int*>(Kbuf + slot * KSLOT + BN * KLD) + gtid : dump_i + gtid;
  };
  auto list_at = [&](int i) __attribute__((always_inline)) {               // list entry i as scalars
    const int4 e = tlist[i];
    return make_int4(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y),
                     __builtin_amdgcn_readfirstlane(e.z), __builtin_amdgcn_readfirstlane(e.w));
  };

  // ---- the three pieces of per-tile work of one wave ------------------------------------------------------
  auto qk = [&](int4 e, int slot) __attribute__((always_inline)) {                       // S^T = K Q^T (+ document / causal mask on edge tiles)
    constexpr bool stage = true;                          // (the tile loads of the next stage ride in its gaps)
    const int jt = e.x, kmin = e.y, kmax = e.z, kminpos = e.w;
    const int k0 = jt * BN;
    act_s = uniform(k0 <= wq0 + 31 && tile_may_interact(wminpos, wmax, kminpos, kmax));
    if (!act_s) {
      if (stage) {
#pragma unroll
        for (int i = 0; i < 5; ++i) issue_piece(i);
      }
      return;
    }
    const bf16_t* Ks = Kbuf + slot * KSLOT;
    // operand ring: QRING fragments in flight before the first MFMA, each MFMA's register refilled with the
    // fragment QRING steps ahead; consecutive MFMAs alternate between the two 32-row K blocks (accumulators).
    // (The compiler's own schedule keeps 1-2 reads ahead, which leaves the phase LDS-latency-bound.)
    const bf16_t* kp = Ks + l31 * KLD + 8 * hi;
    constexpr int NQK = 2 * KSTEPS, QRING = NQK < 8 ? NQK : 8;   // op i: K block i&1, contraction step i>>1
    uint4 kf[QRING];
    const uint4 dflt = make_uint4(lane, tid, l31, hi);
#pragma unroll
    for (int i = 0; i < QRING; ++i) kf[i] = PP_LDS16(kp + (i & 1) * 32 * KLD + 16 * (i >> 1), dflt);
    __builtin_amdgcn_sched_barrier(0);      // (pins the order: the scheduler otherwise sinks the reads again)
#pragma unroll
    for (int i = 0; i < NQK; ++i) {
      sacc[i & 1] = PP_MFMA(kf[i % QRING], qreg[i >> 1], i < 2 ? zero16 : sacc[i & 1]);
      if (i + QRING < NQK) kf[i % QRING] = PP_LDS16(kp + ((i + QRING) & 1) * 32 * KLD + 16 * ((i + QRING) >> 1), dflt);
      if (stage && i >= 1 && i <= 5) issue_piece(i - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool need_mask =
        uniform(!(kmin == kmax && kmax == wminpos && wminpos == wmax && !w_has_zero && (k0 + BN - 1 <= wq0)));
    if (need_mask) {
      const int* docs = reinterpret_cast<const int*>(Ks + BN * KLD);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int4 dk = *reinterpret_cast<const int4*>(docs + 32 * blk + 8 * r4 + 4 * hi);
          const int dkk[4] = {dk.x, dk.y, dk.z, dk.w};
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int kv = k0 + 32 * blk + 8 * r4 + 4 * hi + e2;
            const bool ok = (kv <= qrow) & (dkk[e2] == dq) & (dq > 0);
            sacc[blk][4 * r4 + e2] = ok ? sacc[blk][4 * r4 + e2] : -INFINITY;
          }
        }
      }
    }
  };
  auto softmax = [&]() __attribute__((always_inline)) {                                  // sacc -> pb (online softmax, lane-local row)
    act_p = act_s;
    if (!act_s) return;
    // four independent v_max3 chains (fmaxf() would canonicalise every MFMA output first: 2x the VALU work)
    float mxs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mxs[c] = max3(sacc[c >> 1][8 * (c & 1) + 0], sacc[c >> 1][8 * (c & 1) + 1], sacc[c >> 1][8 * (c & 1) + 2]);
      mxs[c] = max3(mxs[c], sacc[c >> 1][8 * (c & 1) + 3], sacc[c >> 1][8 * (c & 1) + 4]);
      mxs[c] = max3(mxs[c], sacc[c >> 1][8 * (c & 1) + 5], sacc[c >> 1][8 * (c & 1) + 6]);
    }
    float mx = max3(mxs[0], mxs[1], sacc[0][7]);
    mx = max3(mx, mxs[2], sacc[0][15]);
    mx = max3(mx, mxs[3], sacc[1][7]);
    mx = max3(mx, sacc[1][15], sacc[1][15]);
    mx = half_swap_max(mx) * scale_log2;
    if (uniform(!__all(mx - m_run <= 8.f))) {             // deferred rescale, see attn_fwd.hip
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < DBLK; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    const float neg_m = -m_run;
    float psum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        float p[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          p[e] = fast_exp2(fmaf(sacc[blk][8 * sp + e], scale_log2, neg_m));
          psum[e & 3] += p[e];
        }
        u32x4_t t = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
        pb[blk][sp] = __builtin_bit_cast(bf16x8_t, t);
      }
    }
    l_run += (psum[0] + psum[1]) + (psum[2] + psum[3]);
  };
  auto pv = [&](auto G, auto STAGE, int slot, bf16_t* kbase, bf16_t* vbase, int* dbase) __attribute__((always_inline)) {   // O^T += V^T P^T
    constexpr bool stage = decltype(STAGE)::value;
    if (!act_p) {
      if (stage) {
#pragma unroll
        for (int i = 0; i < 8; ++i) store_piece(G, i, kbase, vbase, dbase);
      }
      return;
    }
    const bf16_t* vp = Vbuf + slot * VSLOT + l31 * VLD + 8 * hi;
    // 4 (16-kv sub-block) x DBLK MFMAs; operand ring of RING fragments, consecutive MFMAs hit different accumulators
    constexpr int NOP = 4 * DBLK, RING = NOP < 8 ? NOP : 8;
    uint4 vf[RING];
    const uint4 dflt = make_uint4(lane, tid, l31, hi);
#pragma unroll
    for (int i = 0; i < RING; ++i) vf[i] = PP_LDS16(vp + (i % DBLK) * 32 * VLD + 16 * (i / DBLK), dflt);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NOP; ++i) {
      const int db = i % DBLK, g = i / DBLK;              // g: 16-kv sub-block = (blk, sp)
      oacc[db] = PP_MFMA(vf[i % RING], pb[g >> 1][g & 1], oacc[db]);
      if (i + RING < NOP) vf[i % RING] = PP_LDS16(vp + ((i + RING) % DBLK) * 32 * VLD + 16 * ((i + RING) / DBLK), dflt);
      if (stage && i < 8) store_piece(G, i, kbase, vbase, dbase);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop ------------------------------------------------------------------------------------------
  if (n > 0) {                              // (uniform over the workgroup)
    // prologue: group A puts K(0) in slot 0 and starts loading V(0); group B puts K(1) in slot 1, loads K(2)
    issue_setup(K + head_off, list_at(grp).x);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_piece(i);
    {
      bf16_t* kb = k_spot(grp);
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(kb + k * KLD) = stg[k];
      *d_spot(grp) = dstage;
    }
    int ps = grp == 0 ? 0 : 2;              // list position this group ISSUES loads for next
    issue_setup(stage_src, list_at(ps).x);
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_piece(i);
    ++ps;
    int4 e_cur = list_at(0);
    int kslot = 0;                          // t % 3
    __syncthreads();

    // Both groups run the SAME instruction stream
    //     [ P.V(t-1) + store pieces , QK^T(t) + issue pieces | barrier | softmax(t) | barrier ];
    // group B enters it one barrier late, which puts its MFMA half beside group A's VALU half and vice versa.
    //   group A: V(t) -> V slot t&1 during interval 2t   (V(t-2) was last read in interval 2t-1; first read 2t+2)
    //   group B: K(t+2) -> K slot (t+2)%3 during 2t+1    (K(t-1) was last read in interval 2t-1; first read 2t+4)
    // The whole loop is instantiated once per group (group as a compile-time constant): a group branch inside
    // the loop gives every accumulator several definitions per iteration and the register allocator spills.
    auto main_loop = [&](auto G) __attribute__((always_inline)) {
      constexpr int grp_c = decltype(G)::value;
      if (grp_c == 1) __syncthreads();
#ifdef TN_PP_TRACE
      const bool trace_on = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (tid & 255) == 0;
#endif
      for (int t = 0; t < n; ++t) {
        PP_STAMP(0);
        const int4 e_nxt = tlist[t + 1];    // (vector read now, scalarised at the end of the iteration)
        const int js = __builtin_amdgcn_readfirstlane(tlist[ps].x);
        const int kst = kslot == 0 ? 2 : kslot - 1;       // (t + 2) % 3
        pv(G, IC<1>{}, (t - 1) & 1, k_spot(kst), v_spot(t & 1), d_spot(kst));
        PP_STAMP(1);
        issue_setup(stage_src, js);
        ++ps;
        qk(e_cur, kslot);
        PP_STAMP(2);
        __syncthreads();
        PP_STAMP(3);
        softmax();
        PP_STAMP(4);
        e_cur = make_int4(__builtin_amdgcn_readfirstlane(e_nxt.x), __builtin_amdgcn_readfirstlane(e_nxt.y),
                          __builtin_amdgcn_readfirstlane(e_nxt.z), __builtin_amdgcn_readfirstlane(e_nxt.w));
        kslot = kslot == 2 ? 0 : kslot + 1;
        PP_STAMP(6);
        __syncthreads();
        PP_STAMP(7);
      }
      pv(G, IC<0>{}, (n - 1) & 1, nullptr, nullptr, nullptr);   // drain: P.V of the last tile
      __syncthreads();
      if (grp_c == 0) __syncthreads();
    };
    if (grp == 0) main_loop(IC<0>{}); else main_loop(IC<1>{});
  }

  // ---- epilogue
#ifdef TN_PP_TRACE
  if (tid == 0 && lin_block < 8192) {
    g_pp_blocks[lin_block][1] = __builtin_amdgcn_s_memrealtime();
    if (lin_block == 8191 || lin_block == gridDim.x * gridDim.y * gridDim.z - 1) {
      g_pp_trace[0][191][2] = __builtin_amdgcn_s_memtime();
      g_pp_trace[0][191][3] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif
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

}  // namespace tn

using namespace tn;

#ifdef TN_PP_TRACE
extern "C" int tn_debug_pp_trace(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tn::g_pp_trace), sizeof(unsigned long long) * 2 * 192 * 8);
}
extern "C" int tn_debug_pp_blocks(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tn::g_pp_blocks), sizeof(unsigned long long) * 8192 * 2);
}
#endif

// called from attn_fwd.hip's launcher
int tn_attn_fwd_pp_launch(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                          AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, int D, float sl2, hipStream_t st) {
  dim3 grid(Nh, qv.tiles(0, 256) + qv.tiles(1, 256), B), block(512);
  if (D == 128)
    hipLaunchKernelGGL((attn_fwd_pp_kernel<128>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)o, lse2, doc, m, qv, T, Nh, Nkv, sl2);
  else if (D == 64)
    hipLaunchKernelGGL((attn_fwd_pp_kernel<64>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)o, lse2, doc, m, qv, T, Nh, Nkv, sl2);
  else
    return TN_EINVAL;
  TN_LAUNCH_CHECK();
  return TN_OK;
}
