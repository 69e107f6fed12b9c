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

// Timing experiments only (scripts/build_variant.sh <name> -DTN_BWD_ABL=3; results are wrong): no MFMA / softmax work,
// the staging skeleton alone.  (Values 1 and 2 — no in-loop loads / no barriers — existed for the register-staged kernels
// this file replaced; their numbers are quoted in the dK/dV kernel's header.)
#ifndef TN_BWD_ABL
#define TN_BWD_ABL 0
#endif
// dK/dV stream: query rows per stage (32 or 64) and ring depth
#ifndef TN_KV_BQ
#define TN_KV_BQ 64
#endif
#ifndef TN_KV_NST
#define TN_KV_NST 2
#endif

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
//
// The (query head of the GQA group, 32-row query stage) pairs a workgroup meets form ONE stream of stages; each stage
// = {Q rows, dO rows, lse, delta, doc ids} travels global -> LDS by LDS-DMA (buffer_load ... lds, no staging
// registers, no LDS store instructions) into a ring of NST slots, NST - 1 stages ahead of the one being computed, with
// counted vmcnt waits and ONE raw barrier per stage.  Why: with register staging the prefetch distance was one tile
// and the per-tile {wait for HBM/L2, store, 2 barriers} skeleton alone took 134 of the dV kernel's 281 us on the
// headline workload (scripts/attn_ktimes.sh, -DTN_BWD_ABL variants of the previous kernel: no loads/stores 224 us, no
// barriers either 206 us, no MFMA/softmax 134 us).  The per-tile document-id statistics come from an LDS window
// filled once per workgroup instead of dependent scalar loads per tile.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct QStage {      // one stage of the stream; every field is wave-uniform (SGPRs)
  int valid;
  int qsb;           // global position of the stage's first query row
  int lrow;          // its row in the local Q / dO / LSE / delta buffers
  int left;          // valid rows (1..32)
  int h;             // query head
  int mn, mx, mp;    // document-id statistics of the 64-position tile the stage lies in
};

template <int D, int MODE>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ Delta,
    bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, const int* __restrict__ doc, AttnMeta meta, QView qv, int T,
    int Nh, int Nkv, float scale, float scale_log2) {
  constexpr bool DO_DV = MODE != 1, DO_DK = MODE != 0;
  constexpr int BNK = 128, BQ = TN_KV_BQ, NST = TN_KV_NST, SPT = kTile / BQ;   // SPT stages per 64-position q tile
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  using Tile = PTile<BQ, D>;
  constexpr int IMGB = Tile::SIZE * 2;          // bytes of one panel image
  constexpr int NPC = Tile::NP * (BQ / 16);     // 1-KiB DMA pieces per image: 16 rows of one panel each
  constexpr int PPW = NPC / 4;                  // pieces per wave and image
  constexpr int IPS = 2 * PPW + 1;              // DMA instructions per wave and stage (Q, dO pieces + one aux row)
  constexpr int STAGEB = 2 * IMGB + 4 * 256;    // {Q image | dO image | lse[64] | delta[64] | doc[64] | spare[64]}
  constexpr int LCAP = 256;                     // stage-list chunk: one candidate stage per thread
  // ONE LDS variable on purpose: with two, hipcc's module-LDS lowering attaches alias scopes to every access and the
  // waitcnt insertion then puts `s_waitcnt vmcnt(0)` in front of the first LDS read that may alias a pending LDS-DMA
  // (= every read of the ring), which serialises the ring (see attn_common.h, i32x4_t).
  // Round 6 (attn_fwd_stream.hip's treatment), D = 128 only: the stored list of query tiles instead of dependent metadata
  // loads; before the ring starts its area holds each wave's private K / V images (the 32 rows arrive by LDS-DMA as
  // 64-byte runs); behind the last stage the dV / dK rows leave through it as whole rows.  Same box, interleaved
  // (profiles/r06c_*): the D = 128 passes gain 1-8 % (most on short documents), the D = 64 pass LOSES 1-4 % (187 -> 228
  // registers for the same two waves per SIMD; its row loads and stores are half as many) and keeps the direct form.
  using KTile = PTile<32, D>;
  constexpr int KIMGB = KTile::SIZE * 2;
  constexpr int OSTR = 2 * D + 16;
  constexpr bool R6 = D == 128;
  constexpr int RING0 = NST * STAGEB, RING1 = 4 * 2 * 32 * OSTR, RING2 = 4 * 2 * KIMGB;
  constexpr int RINGX = RING0 > RING1 ? (RING0 > RING2 ? RING0 : RING2) : (RING1 > RING2 ? RING1 : RING2);
  constexpr int RING = D == 128 ? RINGX : RING0;
  __shared__ __attribute__((aligned(1024))) char smem[RING + (LCAP + NST) * 32 + 16 + kListPre * 16];
  i32x4_t* slist = reinterpret_cast<i32x4_t*>(smem + RING);             // entry e = {slist[2e], slist[2e + 1]}
  int* wcount = reinterpret_cast<int*>(smem + RING + (LCAP + NST) * 32);
  i32x4_t* qent = reinterpret_cast<i32x4_t*>(smem + RING + (LCAP + NST) * 32 + 16);   // the stored q-tile list

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int kt = blockIdx.y, hk = blockIdx.x, b = blockIdx.z;   // (see head_of_slot: head in x, heavy tiles first)
  const int G = Nh / Nkv;
  const int k0 = kt * BNK;
  const int wk0 = k0 + 32 * wave;
  const int kvrow = wk0 + l31;
  const bool kvalid = kvrow < T;

  const bool bidir = qv.bidir != 0;
  // ---- round trip A: the stored list of query tiles, the id statistics of this wave's rows, this lane's id, and the
  // wave's K / V rows — issued together
  const bool plain = R6 && qv.nseg == 1 && qv.off[0] == 0 && qv.row0[0] == 0 && !bidir;   // (the stored lists are causal)
  i32x4_t ql_head = {kListPre + 1, 0, 0, 0}, ql_mine = {0, 0, 0, 0};
  if (plain) {
    const i32x4_t* ql = reinterpret_cast<const i32x4_t*>(meta.qlist) + ((size_t)b * meta.nq128 + kt) * (1 + kListPre);
    ql_head = ql[0];
    if (tid < kListPre) ql_mine = ql[1 + tid];
  }
  i32x4_t ws4 = {0x7fffffff, 0, 1, 0};          // {min positive id, max id, pad present, -} of the wave's 32 kv rows
  if (R6 && wk0 < T) ws4 = reinterpret_cast<const i32x4_t*>(meta.qstat)[(size_t)b * meta.nq32 + wk0 / 32];
  const int dkdoc = kvalid ? doc[(size_t)b * T + kvrow] : 0;
  bf16x8_t kreg[KSTEPS], vreg[DO_DK ? KSTEPS : 1];
  if constexpr (!R6) {
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
  } else {
    const size_t krow_elems0 = (size_t)Nkv * D;
    const uint32_t k_bytes0 = (uint32_t)min((size_t)T * krow_elems0 * 2, (size_t)0x7fffffff);
    const __amdgpu_buffer_rsrc_t rk0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)b * T * krow_elems0), 0, k_bytes0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)b * T * krow_elems0), 0, k_bytes0, 0x00020000);
    const int rrk = lane >> 2;
    const uint32_t voffk = (uint32_t)(((size_t)rrk * krow_elems0 + 8 * ((lane & 3) ^ ((rrk >> 2) & 3))) * 2);
    char* kpriv = smem + wave * (2 * KIMGB);    // {K image | V image} of this wave's 32 rows
#pragma unroll
    for (int pc = 0; pc < KTile::NP * 2; ++pc) {
      const int panel = pc % KTile::NP, rh = pc / KTile::NP;
      const uint32_t vo = (wk0 + 16 * rh + rrk < T) ? voffk : 0x80000000u;
      const uint32_t so = (uint32_t)((((size_t)wk0 + 16 * rh) * krow_elems0 + (size_t)hk * D + 32 * panel) * 2);
      char* dst = kpriv + panel * (KTile::PSTRIDE * 2) + rh * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk0, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      if (DO_DK) __builtin_amdgcn_raw_ptr_buffer_load_lds(rv0, (lds_ptr_t)(dst + KIMGB), 16, vo, so, 0, 0);
    }
    wait_vmcnt<0>();
    asm volatile("" ::"v"(dkdoc));             // (hipcc's own wait for this load must sit HERE, not inside the stage loop)
    const PRowReader<32, D> krd(l31, hi);
    const bf16_t* ki = reinterpret_cast<const bf16_t*>(smem + wave * (2 * KIMGB));
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      kreg[s] = krd.operand(ki, 0, s);
      if (DO_DK) vreg[s] = krd.operand(ki + KTile::SIZE, 0, s);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kreg[s]));
      if (DO_DK) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vreg[s]));
    }
  }
  const int n_pre = __builtin_amdgcn_readfirstlane(ql_head.x);
  const bool pre = n_pre <= kListPre;          // the stored list is complete: use it
  if (pre && tid < n_pre) qent[tid] = ql_mine;
  // everybody has read its private images (the ring area is free) and the list is in LDS
  __syncthreads();

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = 2 * kt, t1 = min(2 * kt + 1, meta.nt - 1);
  const int kvcap = bidir ? -0x7fffffff : kvrow;            // `kvcap <= q`: the causal term of the predicate
  int bminpos = 0, bmax = 0, qt_lo = k0 / kTile, qt_end = 0;
  if (!pre) {
    // first 64-position query tile, global index: q >= kv under the causal mask; bidirectional: the first tile that
    // shares a document with these kv rows
    if (bidir) qt_lo = min(meta.q_lo[(size_t)b * meta.nt + t0], meta.q_lo[(size_t)b * meta.nt + t1]);
    bminpos = min(m_minpos[t0], m_minpos[t1]);
    bmax = max(m_max[t0], m_max[t1]);
    const int qhi64 = max(meta.kv_hi[(size_t)b * meta.nt + t0], meta.kv_hi[(size_t)b * meta.nt + t1]);
    qt_end = min(qhi64 + 1, meta.nt);                       // exclusive
  }
  int wminpos, wmax;
  bool w_uniform;                                           // all 32 kv rows in one document
  if constexpr (R6) {
    const int4 wsc = scalarize(ws4);
    wminpos = wsc.x;
    wmax = wsc.y;
    w_uniform = wminpos == wmax && wsc.z == 0;
  } else {
    wave_id_range(dkdoc, wminpos, wmax);
    w_uniform = (wminpos == wmax) && !__any(dkdoc == 0);
  }
  // query tiles this launch owns: per segment, the global 64-tile range clipped to [qt_lo, qt_end)
  int seg_lo[2], seg_n[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int first = qv.off[s] / kTile, cnt = qv.tiles(s, kTile);
    seg_lo[s] = max(qt_lo, first);
    seg_n[s] = max(min(qt_end, first + cnt) - seg_lo[s], 0);
  }
  const int nqt = pre ? n_pre : seg_n[0] + seg_n[1];
  // ---- the stream of stages: (head in group, BQ-row part of a 64-row q tile), skipping what cannot interact.
  // The per-stage bookkeeping used to be a scalar scan every wave ran between two stages (≈100 SALU instructions per
  // stage and wave: 7 SALU per MFMA in the dV kernel's PMC); now the workgroup compacts LCAP candidate stages at a time
  // into an LDS list — one candidate per thread, ballot compaction as in attn_common.h — and the loop reads entries.
  const int per_head = SPT * nqt, total_c = per_head * G;
  auto build_list = [&](int cb) {       // candidates [cb, cb + LCAP) -> n entries (+ NST invalid ones behind them)
    const int c = cb + tid;
    QStage d = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < total_c) {
      const int g = c / per_head, r = c - g * per_head;
      const int idx = r / SPT, part = r % SPT;
      int sg = 0, t64, lt, left;
      if (pre) {                                  // (plain launch: local row = global position)
        const i32x4_t qe = qent[idx];
        t64 = qe.x; d.mn = qe.y; d.mx = qe.z; d.mp = qe.w;
        lt = t64;
        left = T - t64 * kTile - BQ * part;
        d.valid = left > 0;
      } else {
        sg = idx >= seg_n[0] ? 1 : 0;
        t64 = seg_lo[sg] + idx - (sg ? seg_n[0] : 0);
        lt = t64 - qv.off[sg] / kTile;
        left = min(qv.rows[sg] - lt * kTile, T - t64 * kTile) - BQ * part;
        d.mp = m_minpos[t64];
        d.mx = m_max[t64];
        d.mn = m_min[t64];
        d.valid = left > 0 && tile_may_interact(d.mp, d.mx, bminpos, bmax);
      }
      d.qsb = t64 * kTile + BQ * part;
      d.lrow = qv.row0[sg] + lt * kTile + BQ * part;
      d.left = min(left, BQ);
      d.h = hk * G + g;
    }
    const int wv = tid >> 6;
    const unsigned long long bal = __ballot(d.valid != 0);
    if (lane == 0) wcount[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int cnt = wcount[w];
      before += w < wv ? cnt : 0;
      total += cnt;
    }
    if (d.valid) {
      const int e = before + __popcll(bal & ((1ull << lane) - 1ull));
      slist[2 * e] = i32x4_t{d.qsb, d.lrow, d.left, d.h};
      slist[2 * e + 1] = i32x4_t{d.mp, d.mx, d.mn, 1};
    }
    const int n = __builtin_amdgcn_readfirstlane(total);
    if (tid < NST) {                    // what the ring reads past the end: stages that load nothing
      slist[2 * (n + tid)] = i32x4_t{0, 0, 0, 0};
      slist[2 * (n + tid) + 1] = i32x4_t{0, 0, 0, 0};
    }
    __syncthreads();
    return n;
  };
  auto entry = [&](int e) {
    const int4 a = scalarize(slist[2 * e]), c = scalarize(slist[2 * e + 1]);
    QStage d = {c.w, a.x, a.y, a.z, a.w, c.z, c.y, c.x};
    return d;
  };

  // ---- LDS-DMA sources: descriptors over this batch row's slices, lane part of the offsets
  const size_t qrow_elems = (size_t)Nh * D;
  const uint32_t q_bytes = (uint32_t)min((size_t)qv.rpb * qrow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)(dO + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const uint32_t s_bytes = (uint32_t)((size_t)Nh * qv.rpb * 4);
  const __amdgpu_buffer_rsrc_t rlse =
      __builtin_amdgcn_make_buffer_rsrc((void*)(LSE2 + (size_t)b * Nh * qv.rpb), 0, s_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdelta =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Delta + (size_t)b * Nh * qv.rpb), 0, s_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdoc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(doc + (size_t)b * T), 0, (uint32_t)T * 4, 0x00020000);
  // lane L of a piece writes LDS chunk L = (row L >> 2, physical chunk L & 3) of a 16-row x 64-byte panel slab
  const int rr = lane >> 2;
  const uint32_t voff = (uint32_t)(((size_t)rr * qrow_elems + 8 * ((lane & 3) ^ ((rr >> 2) & 3))) * 2);
  constexpr uint32_t OOB = 0x80000000u;       // >= num_records: the load returns 0 and touches no memory
  // Always IPS instructions (the vmcnt arithmetic of the ring stays uniform): an invalid stage has left = 0, every
  // lane is out of range, nothing is read and its slot is filled with zeros that nobody looks at.
  auto issue = [&](const QStage& d, int slot) {
    char* st = smem + slot * STAGEB;
    const uint32_t base = (uint32_t)(((size_t)d.lrow * Nh + d.h) * D * 2);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 4 * i, panel = pc % Tile::NP, rh = pc / Tile::NP;
      const uint32_t vo = (16 * rh + rr < d.left) ? voff : OOB;
      const uint32_t so = base + (uint32_t)((16 * rh * qrow_elems + 32 * panel) * 2);
      char* dst = st + panel * (Tile::PSTRIDE * 2) + rh * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdo, (lds_ptr_t)(dst + IMGB), 16, vo, so, 0, 0);
    }
    const uint32_t va = lane < d.left ? (uint32_t)lane * 4 : OOB;
    char* aux = st + 2 * IMGB;
    const uint32_t srow = (uint32_t)(((size_t)d.h * qv.rpb + d.lrow) * 4);
    if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rlse, (lds_ptr_t)aux, 4, va, srow, 0, 0);
    else if (wave == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdelta, (lds_ptr_t)(aux + 256), 4, va, srow, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rdoc, (lds_ptr_t)(aux + 256 * wave), 4, va, (uint32_t)d.qsb * 4, 0, 0);
  };

  f32x16_t dkacc[DBLK], dvacc[DBLK];     // (the unused one of a single-output MODE is dead code)
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (DO_DK) dkacc[i][r] = 0.f;
      if (DO_DV) dvacc[i][r] = 0.f;
    }

  const PRowReader<BQ, D> rrd(l31, hi);
  const PTrReader<BQ, D> trd(lane);
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  int slot = 0;
  for (int cb = 0; cb < total_c; cb += LCAP) {
   const int n = build_list(cb);
   // ring[0] = the stage being computed, ring[1 .. NST - 2] = the stages in flight behind it
   QStage ring[NST - 1];
#pragma unroll
   for (int i = 0; i < NST - 1; ++i) {
     ring[i] = entry(i);
     issue(ring[i], (slot + i) % NST);
   }
   for (int it = 0; it < n; ++it) {
    const QStage cur = ring[0];
    const QStage ahead = entry(it + NST - 1);    // (LDS read, off the critical path: before the wait)
    // my pieces of `cur` have landed (the NST - 2 stages after it may stay in flight) ...
    wait_vmcnt<(NST - 2) * IPS>();
    // ... everybody's have, and everybody has left the previous stage: its slot takes the stage NST - 1 ahead
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(ahead, (slot + NST - 1) % NST);

    const bf16_t* Qs = reinterpret_cast<const bf16_t*>(smem + slot * STAGEB);
    const bf16_t* dOs = Qs + Tile::SIZE;
    const float* lse_s = reinterpret_cast<const float*>(smem + slot * STAGEB + 2 * IMGB);
    const float* delta_s = lse_s + 64;
    const int* docq = reinterpret_cast<const int*>(lse_s + 128);
    if (uniform(TN_BWD_ABL != 3 && (bidir || cur.qsb + BQ - 1 >= wk0) && tile_may_interact(cur.mp, cur.mx, wminpos, wmax))) {
      const bool q_uniform = w_uniform && cur.mn == cur.mx && cur.mx == wmax && cur.left == BQ;
      // The work on one 32-row half (qs) of the stage, in pieces.  (Issuing both halves of an interior stage as ONE
      // straight-line block — S of half 1 under the exponentials of half 0 — measured 3-9 % SLOWER: 226 -> 242 VGPRs.)
      auto s_of = [&](int qs) {          // S[q, kv] = Q K^T: rows = q in registers, column = this lane's kv
        f32x16_t a = mfma32(rrd.operand(Qs, 32 * qs, 0), kreg[0], zero16);
#pragma unroll
        for (int s = 1; s < KSTEPS; ++s) a = mfma32(rrd.operand(Qs, 32 * qs, s), kreg[s], a);
        return a;
      };
      auto dp_of = [&](int qs) {         // dP[q, kv] = dO V^T
        f32x16_t a = mfma32(rrd.operand(dOs, 32 * qs, 0), vreg[0], zero16);
#pragma unroll
        for (int s = 1; s < KSTEPS; ++s) a = mfma32(rrd.operand(dOs, 32 * qs, s), vreg[DO_DK ? s : 0], a);
        return a;
      };
      auto probs = [&](int qs, const f32x16_t& sacc, float (&p)[16], auto masked) {
        constexpr bool MASK = decltype(masked)::value;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int o = 32 * qs + 8 * r4 + 4 * hi;
          const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + o);
          const float le[4] = {l4.x, l4.y, l4.z, l4.w};
          int qd[4] = {0, 0, 0, 0};
          if (MASK) {
            const i32x4_t q4 = *reinterpret_cast<const i32x4_t*>(docq + o);
            qd[0] = q4.x; qd[1] = q4.y; qd[2] = q4.z; qd[3] = q4.w;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float pv = fast_exp2(sacc[4 * r4 + e] * scale_log2 - le[e]);
            if (MASK) pv = ((kvcap <= cur.qsb + o + e) & (qd[e] == dkdoc) & (dkdoc > 0)) ? pv : 0.f;
            p[4 * r4 + e] = pv;
          }
        }
      };
      auto packed = [&](const float (&p)[16], int sp) {
        const u32x4_t t = {pack2bf(p[8 * sp + 0], p[8 * sp + 1]), pack2bf(p[8 * sp + 2], p[8 * sp + 3]),
                           pack2bf(p[8 * sp + 4], p[8 * sp + 5]), pack2bf(p[8 * sp + 6], p[8 * sp + 7])};
        return __builtin_bit_cast(bf16x8_t, t);
      };
      // acc += IMG^T[d, q] X[q, kv] over the 32 q rows of half qs (X = P or dS in registers, IMG^T by transpose reads)
      auto acc_tr = [&](int qs, const bf16_t* img, f32x16_t (&acc)[DBLK], const float (&p)[16]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const bf16x8_t x = packed(p, sp);
#pragma unroll
          for (int db = 0; db < DBLK; ++db) acc[db] = mfma32(trd.operand(img, db, 32 * qs + 16 * sp), x, acc[db]);
        }
      };
      auto to_ds = [&](int qs, float (&p)[16], const f32x16_t& dpacc) {   // p becomes dS = P o (dP - delta)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(delta_s + 32 * qs + 8 * r4 + 4 * hi);
          const float de[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) p[4 * r4 + e] *= dpacc[4 * r4 + e] - de[e];
        }
      };
      // (Round 6 tried the operand reads of a half as inline-asm batches with counted lgkmcnt waits, the form that serves
      //  attn_fwd_stream.hip: 1.5 % SLOWER here on both D = 64 shapes, same box — profiles/r06c_*; hipcc's own reads stay.)
#pragma unroll
      for (int qs = 0; qs < BQ / 32; ++qs) {
        const int qsb = cur.qsb + 32 * qs;
        if (!uniform(bidir || qsb + 31 >= wk0)) continue;       // every q of this half precedes the kv rows
        const bool need_mask = uniform(!(q_uniform && (bidir || qsb >= wk0 + 31)));
        const f32x16_t sacc = s_of(qs);
        f32x16_t dpacc = zero16;
        if (DO_DK) dpacc = dp_of(qs);
        float p[16];
        if (need_mask) probs(qs, sacc, p, std::true_type{}); else probs(qs, sacc, p, std::false_type{});
        if (DO_DV) acc_tr(qs, dOs, dvacc, p);
        if (DO_DK) {
          to_ds(qs, p, dpacc);
          acc_tr(qs, Qs, dkacc, p);
        }
      }
    }
#pragma unroll
    for (int i = 0; i + 1 < NST - 1; ++i) ring[i] = ring[i + 1];
    ring[NST - 2] = ahead;
    slot = (slot + 1) % NST;
   }
   wait_vmcnt<0>();      // (the zero-fill tail DMAs must not land on the next chunk's stages)
   __syncthreads();
  }

  // ---- epilogue: the ring is quiet behind the last chunk's barrier; each wave writes its 32 x D blocks (8-byte runs of
  // the accumulator layout) into private images and reads them back as whole rows — 16-byte stores, 64 / (D / 8) rows per
  // instruction, instead of 8 bytes per lane at a row stride (attn_fwd_stream.hip)
  if constexpr (!R6) {
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
  } else {
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
    char* obv = smem + wave * (2 * 32 * OSTR);
    char* obk = obv + 32 * OSTR;
#pragma unroll
    for (int db = 0; db < DBLK; ++db) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        if (DO_DV) {
          const u32x2_t o = {pack2bf(dvacc[db][4 * r4 + 0], dvacc[db][4 * r4 + 1]),
                             pack2bf(dvacc[db][4 * r4 + 2], dvacc[db][4 * r4 + 3])};
          *reinterpret_cast<u32x2_t*>(obv + l31 * OSTR + (32 * db + 8 * r4 + 4 * hi) * 2) = o;
        }
        if (DO_DK) {
          const u32x2_t o = {pack2bf(dkacc[db][4 * r4 + 0] * scale, dkacc[db][4 * r4 + 1] * scale),
                             pack2bf(dkacc[db][4 * r4 + 2] * scale, dkacc[db][4 * r4 + 3] * scale)};
          *reinterpret_cast<u32x2_t*>(obk + l31 * OSTR + (32 * db + 8 * r4 + 4 * hi) * 2) = o;
        }
      }
    }
    constexpr int CPR = D / 8, RPI = 64 / CPR;           // 16-byte chunks per row, rows per store instruction
    const int cc = lane % CPR, r0 = lane / CPR;
    const size_t off = (((size_t)b * T + wk0) * Nkv + hk) * D + cc * 8;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int row = i * RPI + r0;
      u32x4_t v4 = {0, 0, 0, 0}, k4 = {0, 0, 0, 0};
      if (DO_DV) v4 = *reinterpret_cast<const u32x4_t*>(obv + row * OSTR + cc * 16);
      if (DO_DK) k4 = *reinterpret_cast<const u32x4_t*>(obk + row * OSTR + cc * 16);
      if (wk0 + row < T) {
        if (DO_DV) *reinterpret_cast<u32x4_t*>(dV + off + (size_t)row * Nkv * D) = v4;
        if (DO_DK) *reinterpret_cast<u32x4_t*>(dK + off + (size_t)row * Nkv * D) = k4;
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
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, float* __restrict__ Delta,
    bf16_t* __restrict__ dQ, const int* __restrict__ doc, AttnMeta meta, QView qv, int T, int Nh, int Nkv,
    float scale, float scale_log2, const bf16_t* __restrict__ O) {
  // O != null: this kernel ALSO forms delta = rowsum(dO o O) of its 128 query rows — it holds the dO rows in registers
  // anyway — and writes it to `Delta` for the dK / dV pass, which is launched BEHIND it (round 5: the separate
  // attn_delta_kernel pass, 64 launches and 2.5 ms per Qwen2-Audio step, is gone).  O == null: `Delta` is read.
  constexpr int BM = 128, BN = 64, NST = 2;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  using Tile = PTile<BN, D>;
  constexpr int IMGB = Tile::SIZE * 2;          // bytes of one panel image
  constexpr int NPC = Tile::NP * (BN / 16);     // 1-KiB DMA pieces per image
  constexpr int PPW = NPC / 4;                  // pieces per wave and image
  constexpr int IPS = 2 * PPW + 1;              // DMA instructions per wave and stage
  constexpr int STAGEB = 2 * IMGB + 4 * 256;    // {K image | V image | doc ids[64] | 3 spare rows}
  constexpr int WIN = 256;                      // kv tiles whose statistics are kept in LDS
  // K image: rows -> A of S^T = K Q^T, transposed -> A of dQ^T += K^T dS^T;  V image: rows -> A of dP^T = V dO^T.
  // Stages = the kv tiles this workgroup meets, streamed by LDS-DMA exactly like the dK/dV kernel's (one LDS variable,
  // see there).
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGEB + WIN * 16];
  i32x4_t* kstat = reinterpret_cast<i32x4_t*>(smem + NST * STAGEB);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  float dsum = 0.f;                         // this lane's share of rowsum(dO o O): slots 8 hi .. 8 hi + 7 of every k-step
  {
    const size_t off = (((size_t)b * qv.rpb + (qvalid ? lrow : 0)) * Nh + h) * D + 8 * hi;
    uint4 orow[KSTEPS];                     // (all loads of the row are issued before the first use: ONE memory round trip)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      uint4 a = make_uint4(0, 0, 0, 0), c = make_uint4(0, 0, 0, 0);
      orow[s] = make_uint4(0, 0, 0, 0);
      if (qvalid) {
        a = *reinterpret_cast<const uint4*>(Q + off + 16 * s);
        c = *reinterpret_cast<const uint4*>(dO + off + 16 * s);
        if (O != nullptr) orow[s] = *reinterpret_cast<const uint4*>(O + off + 16 * s);
      }
      qreg[s] = as_bf16x8(a);
      doreg[s] = as_bf16x8(c);
    }
    if (O != nullptr) {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        Vec16<bf16_t> ov, gv;
        float of[8], gf[8];
        ov.raw = orow[s];
        const u32x4_t g4 = __builtin_bit_cast(u32x4_t, doreg[s]);
        gv.raw = make_uint4(g4.x, g4.y, g4.z, g4.w);
        ov.unpack(of);
        gv.unpack(gf);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += of[e] * gf[e];
      }
    }
  }
  const int dq = qvalid ? doc[(size_t)b * T + qrow] : 0;
  const float lse2 = qvalid ? LSE2[((size_t)b * Nh + h) * qv.rpb + lrow] : INFINITY;
  float delta;
  if (O != nullptr) {
    delta = dsum + __shfl_xor(dsum, 32, 64);            // (both 32-lane halves hold the same rows)
    if (qvalid && hi == 0) Delta[((size_t)b * Nh + h) * qv.rpb + lrow] = delta;
  } else {
    delta = qvalid ? Delta[((size_t)b * Nh + h) * qv.rpb + lrow] : 0.f;
  }

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = q0 / kTile, t1 = min(t0 + 1, meta.nt - 1);
  const bool bidir = qv.bidir != 0;
  const int j_hi = bidir ? max(t1, max(meta.kv_hi[(size_t)b * meta.nt + t0], meta.kv_hi[(size_t)b * meta.nt + t1])) : t1;
  const int qcap = bidir ? 0x7fffffff : qrow;               // `kv <= qcap`: the causal term of the predicate
  // statistics window: the WIN tiles that END at the diagonal are the ones a packed batch needs; a longer reach (plain
  // causal beyond 16 k positions) starts below it and refills the window on the way up
  const int j_lo = min(meta.q_lo[(size_t)b * meta.nt + t0], meta.q_lo[(size_t)b * meta.nt + t1]);
  int win_lo = max(j_lo, 0);
  auto fill_window = [&]() {
    for (int i = tid; i < WIN; i += 256) {
      const int t = win_lo + i;
      if (t < meta.nt) kstat[i] = i32x4_t{m_minpos[t], m_max[t], m_min[t], 0};
    }
  };
  fill_window();
  const int bminpos = min(m_minpos[t0], m_minpos[t1]);
  const int bmax = max(m_max[t0], m_max[t1]);
  int wminpos, wmax;
  wave_id_range(dq, wminpos, wmax);
  const bool w_has_zero = __any(dq == 0);
  __syncthreads();

  struct KStage {
    int valid, j, mn, mx, mp;
  };
  int sc_j = j_lo;
  auto next = [&]() {
    KStage d = {0, 0, 0, 0, 0};
    while (sc_j <= j_hi) {
      const int jj = sc_j++;
      if (jj >= win_lo + WIN) {            // (workgroup-uniform: every wave runs the same scan)
        __syncthreads();
        win_lo = jj;
        fill_window();
        __syncthreads();
      }
      const int4 st = scalarize(kstat[jj - win_lo]);
      if (!tile_may_interact(bminpos, bmax, st.x, st.y)) continue;
      d.valid = 1;
      d.j = jj;
      d.mp = st.x;
      d.mx = st.y;
      d.mn = st.z;
      break;
    }
    return d;
  };

  const size_t krow_elems = (size_t)Nkv * D;
  const uint32_t k_bytes = (uint32_t)min((size_t)T * krow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rk =
      __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv =
      __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdoc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(doc + (size_t)b * T), 0, (uint32_t)T * 4, 0x00020000);
  const int rr = lane >> 2;
  const uint32_t voff = (uint32_t)(((size_t)rr * krow_elems + 8 * ((lane & 3) ^ ((rr >> 2) & 3))) * 2);
  constexpr uint32_t OOB = 0x80000000u;
  auto issue = [&](const KStage& d, int slot) {      // always IPS instructions; an invalid stage reads nothing
    char* st = smem + slot * STAGEB;
    const int k0 = d.j * BN;
    const int left = d.valid ? min(T - k0, BN) : 0;
    const uint32_t base = (uint32_t)(((size_t)k0 * Nkv + hk) * D * 2);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 4 * i, panel = pc % Tile::NP, rh = pc / Tile::NP;
      const uint32_t vo = (16 * rh + rr < left) ? voff : OOB;
      const uint32_t so = base + (uint32_t)((16 * rh * krow_elems + 32 * panel) * 2);
      char* dst = st + panel * (Tile::PSTRIDE * 2) + rh * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + IMGB), 16, vo, so, 0, 0);
    }
    const uint32_t va = lane < left ? (uint32_t)lane * 4 : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rdoc, (lds_ptr_t)(st + 2 * IMGB + 256 * wave), 4, va, (uint32_t)k0 * 4, 0,
                                             0);
  };

  f32x16_t dqacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  const PRowReader<BN, D> rrd(l31, hi);
  const PTrReader<BN, D> trd(lane);
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  KStage ring[NST - 1];
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) {
    ring[i] = next();
    issue(ring[i], i);
  }
  int slot = 0;
  while (ring[0].valid) {
    const KStage cur = ring[0];
    const KStage ahead = next();
    wait_vmcnt<(NST - 2) * IPS>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(ahead, (slot + NST - 1) % NST);

    const bf16_t* Ks = reinterpret_cast<const bf16_t*>(smem + slot * STAGEB);
    const bf16_t* Vs = Ks + Tile::SIZE;
    const int* docs = reinterpret_cast<const int*>(smem + slot * STAGEB + 2 * IMGB);
    const int k0 = cur.j * BN;
    if (uniform(TN_BWD_ABL != 3 && (bidir || k0 <= wq0 + 31) && tile_may_interact(wminpos, wmax, cur.mp, cur.mx))) {
      const bool need_mask = uniform(!(cur.mn == cur.mx && cur.mx == wminpos && wminpos == wmax && !w_has_zero &&
                                       (bidir || k0 + BN - 1 <= wq0)));
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        if (uniform(bidir || k0 + 32 * blk <= wq0 + 31)) {    // else: this 32-row KV block is above the diagonal
          f32x16_t sacc = mfma32(rrd.operand(Ks, 32 * blk, 0), qreg[0], zero16);
          f32x16_t dpacc = mfma32(rrd.operand(Vs, 32 * blk, 0), doreg[0], zero16);
#pragma unroll
          for (int s = 1; s < KSTEPS; ++s) {
            sacc = mfma32(rrd.operand(Ks, 32 * blk, s), qreg[s], sacc);
            dpacc = mfma32(rrd.operand(Vs, 32 * blk, s), doreg[s], dpacc);
          }
          float ds[16];
          auto dscore = [&](auto masked) {
            constexpr bool MASK = decltype(masked)::value;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              int dkk[4] = {0, 0, 0, 0};
              if (MASK) {
                const i32x4_t dk = *reinterpret_cast<const i32x4_t*>(docs + 32 * blk + 8 * r4 + 4 * hi);
                dkk[0] = dk.x; dkk[1] = dk.y; dkk[2] = dk.z; dkk[3] = dk.w;
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = 4 * r4 + e;
                float pv = fast_exp2(sacc[r] * scale_log2 - lse2);
                if (MASK) {
                  const int kv = k0 + 32 * blk + 8 * r4 + 4 * hi + e;
                  pv = ((kv <= qcap) & (dkk[e] == dq) & (dq > 0)) ? pv : 0.f;
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
              dqacc[db] = mfma32(trd.operand(Ks, db, 32 * blk + 16 * sp), dsb, dqacc[db]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i + 1 < NST - 1; ++i) ring[i] = ring[i + 1];
    ring[NST - 2] = ahead;
    slot = (slot + 1) % NST;
  }
  wait_vmcnt<0>();      // (the zero-fill tail DMAs)

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

namespace tn {
// attn_bwd_dq_stream.hip: the dQ pass on precomputed tile lists, Q / dO by LDS-DMA, batched operand reads, whole-row stores
void launch_attn_bwd_dq_stream(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO, const float* lse2,
                               float* delta, bf16_t* dQ, const int* doc, AttnMeta m, QView qv, int B, int T, int Nh,
                               int Nkv, int D, float scale, float sl2, const bf16_t* O, const bf16_t* rcos,
                               const bf16_t* rsin, hipStream_t st);
// attn_bwd_fused.hip: dK and dV in ONE pass (D = 128)
void launch_attn_bwd_kv_fused128(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                 const float* lse2, const float* delta, bf16_t* dK, bf16_t* dV, const int* doc,
                                 AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, float scale, float sl2,
                                 const bf16_t* rcos, const bf16_t* rsin, hipStream_t st);
}  // namespace tn

using namespace tn;

extern "C" {

int tn_rope_apply(const void* q, const void* k, void* q_out, void* k_out, const void* cos_t, const void* sin_t, int n,
                  int hq, int hk, int D, int backward, int dtype, void* stream);      // (norm_act.hip)

// TN_ATTN_BWD_KV=split restores the two-launch dV / dK scheme for D = 128 (kernel-development A/B switch, read per call)
static bool bwd_kv_split() {
  const char* e = getenv("TN_ATTN_BWD_KV");
  return e != nullptr && e[0] == 's';
}

// dQ kernel selection: 1 = attn_bwd_dq_stream.hip (default), 0 = this file's kernel (the reference the stream kernel is
// compared against).  TN_ATTN_BWD_DQ = 0 / 1 forces one; tn_attn_set_bwd_dq (development entry point) overrides per process.
static int g_bwd_dq_override = -1;
static int bwd_dq_mode() {
  static int mode = [] {
    const char* e = getenv("TN_ATTN_BWD_DQ");
    return e ? atoi(e) : 1;
  }();
  return g_bwd_dq_override >= 0 ? g_bwd_dq_override : mode;
}

static int attn_bwd_launch(const void* q, const void* k, const void* v, const void* o, const void* dout,
                           const float* lse2, float* delta, void* dq, void* dk, void* dv, const int* doc,
                           const int* meta, int B, int T, int Nh, int Nkv, int D, float scale, QView qv,
                           void* stream, const void* rope_cos = nullptr, const void* rope_sin = nullptr) {
  if (B <= 0 || T <= 0 || Nh <= 0 || Nkv <= 0 || Nh % Nkv) return TN_EINVAL;
  if (D != 64 && D != 128) return TN_EINVAL;
  for (int s = 0; s < qv.nseg; ++s)
    if (qv.off[s] % 128 || qv.row0[s] % 128 || (s + 1 < qv.nseg && qv.rows[s] % 128)) return TN_EINVAL;
  const AttnMeta m = make_attn_meta(meta, B, T);
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.4426950408889634f;
  const size_t rows = (size_t)B * qv.rpb * Nh;
  dim3 gq(Nh, qv.tiles(0, 128) + qv.tiles(1, 128), B), gk(Nkv, (T + 127) / 128, B), block(256);
  const bf16_t *Q = (const bf16_t*)q, *K = (const bf16_t*)k, *V = (const bf16_t*)v, *dO = (const bf16_t*)dout;
  // TN_ATTN_DELTA_KERNEL=1: the round-4 order (separate delta pass, dK / dV, dQ) for A/B runs; default: dQ first — it forms
  // delta from the dO / O rows it loads anyway and leaves it for the dK / dV pass behind it
  static const bool delta_pass = [] { const char* e = getenv("TN_ATTN_DELTA_KERNEL"); return e && e[0] == '1'; }();
  const bf16_t* O_ = delta_pass ? nullptr : (const bf16_t*)o;
  (void)rows;
  const bool dq_stream = bwd_dq_mode() == 1;
  // rope_cos / rope_sin (tn_attn_bwd_rope): dq / dk are wanted as gradients of the UN-rotated q / k.  The default D = 128
  // kernels rotate back in their epilogues; every other combination runs as it is and the row kernel follows (same bits).
  const bool rope = rope_cos != nullptr;
  const bool rope_fused = rope && D == 128 && dq_stream && !delta_pass && !(bwd_kv_split() || qv.bidir) && qv.nseg == 1 &&
                          qv.rpb == T;
  const bf16_t* rc = rope_fused ? (const bf16_t*)rope_cos : nullptr;
  const bf16_t* rs = rope_fused ? (const bf16_t*)rope_sin : nullptr;
  auto launch_dq = [&]() {
    if (dq_stream)
      launch_attn_bwd_dq_stream(Q, K, V, dO, lse2, delta, (bf16_t*)dq, doc, m, qv, B, T, Nh, Nkv, D, scale, sl2, O_, rc, rs,
                                st);
    else if (D == 128)
      hipLaunchKernelGGL((attn_bwd_dq_kernel<128>), gq, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dq, doc, m,
                         qv, T, Nh, Nkv, scale, sl2, O_);
    else
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64>), gq, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dq, doc, m,
                         qv, T, Nh, Nkv, scale, sl2, O_);
  };
  if (D == 128) {
    if (delta_pass)
      hipLaunchKernelGGL((attn_delta_kernel<128>), dim3((rows * 16 + 255) / 256), block, 0, st, (const bf16_t*)o, dO,
                         delta, B, qv.rpb, Nh);
    else
      launch_dq();
    if (bwd_kv_split() || qv.bidir) {          // (the fused pass is causal only)
      hipLaunchKernelGGL((attn_bwd_kv_kernel<128, 0>), gk, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dk,
                         (bf16_t*)dv, doc, m, qv, T, Nh, Nkv, scale, sl2);
      hipLaunchKernelGGL((attn_bwd_kv_kernel<128, 1>), gk, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dk,
                         (bf16_t*)dv, doc, m, qv, T, Nh, Nkv, scale, sl2);
    } else {
      launch_attn_bwd_kv_fused128(Q, K, V, dO, lse2, delta, (bf16_t*)dk, (bf16_t*)dv, doc, m, qv, B, T, Nh, Nkv, scale,
                                  sl2, rc, rs, st);
    }
    if (delta_pass) launch_dq();
  } else {
    if (delta_pass)
      hipLaunchKernelGGL((attn_delta_kernel<64>), dim3((rows * 8 + 255) / 256), block, 0, st, (const bf16_t*)o, dO,
                         delta, B, qv.rpb, Nh);
    else
      launch_dq();
    hipLaunchKernelGGL((attn_bwd_kv_kernel<64, 2>), gk, block, 0, st, Q, K, V, dO, lse2, delta, (bf16_t*)dk,
                       (bf16_t*)dv, doc, m, qv, T, Nh, Nkv, scale, sl2);
    if (delta_pass) launch_dq();
  }
  TN_LAUNCH_CHECK();
  if (rope && !rope_fused)
    return tn_rope_apply(dq, dk, dq, dk, rope_cos, rope_sin, B * T, Nh, Nkv, D, 1, 1 /* bf16 */, stream);
  return TN_OK;
}

// Development entry point (NOT part of the C ABI): dQ kernel for A/B runs inside one process (-1 = TN_ATTN_BWD_DQ / default).
int tn_attn_set_bwd_dq(int mode) {
  g_bwd_dq_override = mode;
  return TN_OK;
}

// delta: float [B, Nh, T] scratch (also an output of this call).  dq/dk/dv are fully overwritten.
int tn_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                int Nkv, int D, float scale, void* stream) {
  const QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T, 0, ~0ull};
  return attn_bwd_launch(q, k, v, o, dout, lse2, delta, dq, dk, dv, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

// tn_attn_bwd for q / k that carry a rotary embedding (the projection's epilogue, tn_gemm_bf16_rope, or tn_rope_apply put it
// there): dq / dk come back as the gradients of the UN-rotated projections — what tn_attn_bwd followed by
// tn_rope_apply(dq, dk, backward = 1) returns, bit for bit, without that pass over dq / dk (the reference: the rotary
// embedding's backward in transformers' apply_rotary_pos_emb under autograd, touchnet/models/llama/parallelize_llama.py's
// attention; cos_t / sin_t: the bf16 [B * T, D / 2] tables of tn_rope_table).  D = 128 rotates in the kernels' epilogues.
int tn_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                     float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                     int Nkv, int D, float scale, const void* cos_t, const void* sin_t, void* stream) {
  if (cos_t == nullptr || sin_t == nullptr || (((uintptr_t)cos_t | (uintptr_t)sin_t) & 15)) return TN_EINVAL;
  const QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T, 0, ~0ull};
  return attn_bwd_launch(q, k, v, o, dout, lse2, delta, dq, dk, dv, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream, cos_t,
                         sin_t);
}

// Bidirectional inside a document, see tn_attn_fwd_bidir.
int tn_attn_bwd_bidir(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse2,
                      float* delta, void* dq, void* dk, void* dv, const int* doc, const int* meta, int B, int T, int Nh,
                      int Nkv, int D, float scale, void* stream) {
  QView qv = {1, {0, 0}, {T, 0}, {0, 0}, T, 0, ~0ull};
  qv.bidir = 1;
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
                    {segs[2], nseg > 1 ? segs[5] : 0}, rows_per_batch, 0, ~0ull};
  return attn_bwd_launch(q, k, v, o, dout, lse2, delta, dq, dk, dv, doc, meta, B, T, Nh, Nkv, D, scale, qv, stream);
}

}  // extern "C"
