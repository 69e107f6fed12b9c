// Packed (document-masked) flash attention FORWARD for gfx950 — the product schedule for everything but the longest rows
// (those: attn_fwd_pp.hip): LDS-DMA ring for K / V, nothing a workgroup needs to know computed inside it.
//
// Why (profiles/r05x_attn_fwd_ablation_short_documents.log, profiles/r06a_*): on the headline's ~790-token documents a
// workgroup of attn_fwd.hip meets ~9 KV tiles; a wave's life there was 23 % prologue (four DEPENDENT memory round trips:
// row ids -> tile range -> tile list -> first K/V tile), 8 % epilogue (8 bytes per lane at a row stride: store-issue
// bound) and 68 % tile trips in which every MFMA of the QK^T chain waited for its own LDS read.  Here, per workgroup:
//   * what it has to do comes from the mask metadata, worked out ONCE per batch for all layers and heads
//     (attn_common.h klist / qstat): the list of KV tiles of its query tile and the id statistics of each wave's rows.
//     Two round trips are left: {list, statistics, Q rows} and the first K / V tile;
//   * Q rows arrive by LDS-DMA as 64-byte runs (ring slot 1 is free until the first trip) and are read into the MFMA
//     B-operand registers from there; K / V tiles travel global -> LDS by LDS-DMA (no staging registers, no LDS store
//     instructions) into a two-slot ring, one stage ahead;
//   * K "row" operands are inline-asm reads in batches of four fragments, two batches in flight, retired by counted
//     lgkmcnt waits (D = 128; at D = 64 hipcc's own reads keep the kernel at 125 registers = 4 waves per SIMD, which
//     is worth more there); V^T operands likewise — the transpose-read builtin would also get an `s_waitcnt vmcnt(0)`
//     from hipcc (attn_common.h) and serialise the tile in flight;
//   * exp2 arguments and row sums on PAIRS of scores (v_pk_fma_f32 / v_pk_add_f32);
//   * the O rows leave through the free ring slot as whole rows, 16 bytes per lane, 64 / (D / 8) rows per store
//     instruction.
// Measured on one box, interleaved (profiles/r06a_attn_fwd_stream_vs_base_same_box.log): 286 -> 238 us on the decoder's
// shape, 257 -> 216 us on the audio tower's, 1217 -> 1107 us on plain causal 2 x 8192; same tiles, same MFMA and
// softmax sequence as attn_fwd.hip (outputs equal to the last bf16 digit but for the summation order of the row sums).
// Also measured, and NOT kept: several heads of a query tile per workgroup sharing the list, with the stream crossing
// head seams (1.00 / 1.15 / 1.33 x the time at 2 / 4 / 8 heads: fewer, longer workgroups balance worse over the CUs than
// the shared prologue saves); records sorted heaviest-first (attn_common.h).
//
// Maths, masking rules, layouts and the output contract are those of attn_fwd.hip (reference semantics:
// transformers/integrations/flex_attention.py:190-201,264-340 via touchnet/models/kimi_audio/modeling_kimi_audio.py:582-585
// and touchnet/models/qwen2_audio/__init__.py:190-193).
#include <stdlib.h>

#include "attn_stream.h"

namespace tn {

// TRACE (timing experiments, scripts/r06_attn_trace.py): lane 0 of every wave stamps s_memtime at the marked points into
// trace[(workgroup * 4 + wave) * 64 + event]
template <int D, bool TRACE = false>
__global__ __launch_bounds__(256, D == 64 ? 3 : 2) void attn_fwd_stream_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
    float* __restrict__ LSE2, const int* __restrict__ doc, AttnMeta meta, QView qv, int T, int Nh, int Nkv,
    float scale_log2, unsigned long long* __restrict__ trace) {
  using namespace fstream;
  int tev = 0;
  auto stamp = [&]() {
    if constexpr (TRACE) {
      const unsigned long long t = __builtin_readcyclecounter();
      const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      if ((threadIdx.x & 63) == 0 && tev < 64) trace[((size_t)wg * 4 + (threadIdx.x >> 6)) * 64 + tev] = t;
      ++tev;
    }
  };
  stamp();                                                   // 0: entry
  constexpr int BM = 128, BN = 64, NST = 2;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  using Tile = PTile<BN, D>;
  constexpr int IMGB = Tile::SIZE * 2;          // bytes of one panel image
  constexpr int NPC = Tile::NP * (BN / 16);     // 1-KiB DMA pieces per image: 16 rows of one panel each
  constexpr int PPW = NPC / 4;                  // pieces per wave and image
  constexpr int IPS = 2 * PPW + 1;              // DMA instructions per wave and stage
  constexpr int OSTR = 2 * D + 16;              // row stride (bytes) of the O staging image: conflict-free 8-byte stores
  constexpr int STAGE_KV = 2 * IMGB + 4 * 256;  // {K image | V image | doc ids[64] per wave}
  constexpr int STAGEB = STAGE_KV > BM * OSTR ? STAGE_KV : BM * OSTR;      // a slot also stages 128 O rows
  constexpr int CAP = 192;                      // tile-list chunk
  // ONE LDS variable (attn_bwd.hip explains why two would serialise the DMA ring)
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGEB + (CAP + 4) * 16 + 16];
  i32x4_t* tlist = reinterpret_cast<i32x4_t*>(smem + NST * STAGEB);
  int* wcount = reinterpret_cast<int*>(smem + NST * STAGEB + (CAP + 4) * 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int h = head_of_slot(blockIdx.x, Nh, Nkv), b = blockIdx.z;
  const int hk = h / (Nh / Nkv);
  const bool bidir = qv.bidir != 0;
  int lq0, q0, qleft;                      // local first row / global first position / rows left in the segment
  qv.tile(gridDim.y - 1 - blockIdx.y, BM, lq0, q0, qleft);
  const int wq0 = q0 + 32 * wave;          // GLOBAL position of the wave's first query row
  const int qrow = wq0 + l31;              // global position: what the causal / document predicate compares
  const int lrow = lq0 + 32 * wave + l31;  // row in the local Q / O / LSE buffers
  const bool qvalid = (32 * wave + l31 < qleft) && (qrow < T);

  // ---- round trip A: list of KV tiles, id statistics of the wave's rows, the lane's own id — issued together, in front
  // of the Q rows' DMA
  const bool pre_ok = !bidir && qv.kv_tpc == 0;          // (the stored lists are causal and cover every chunk)
  i32x4_t kl_head = {kListPre + 1, 0, 0, 0}, kl_first = {0, 0, 0, 0}, kl_mine = {0, 0, 0, 0};
  if (pre_ok) {
    const i32x4_t* kl = reinterpret_cast<const i32x4_t*>(meta.klist) +
                        ((size_t)b * meta.nq128 + q0 / BM) * (1 + kListPre);
    kl_head = kl[0];
    kl_first = kl[1];
    if (tid < kListPre) kl_mine = kl[1 + tid];
  }
  i32x4_t qs4 = {0x7fffffff, 0, 1, 0};
  if (wq0 < T) qs4 = reinterpret_cast<const i32x4_t*>(meta.qstat)[(size_t)b * meta.nq32 + wq0 / 32];
  const int dq = qvalid ? doc[(size_t)b * T + qrow] : 0;      // (first used by a masked trip)

  // ---- LDS-DMA sources: descriptors over this batch row's Q / K / V / id slices, lane part of the offsets
  const size_t qrow_elems = (size_t)Nh * D;
  const uint32_t q_bytes = (uint32_t)min((size_t)qv.rpb * qrow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const size_t krow_elems = (size_t)Nkv * D;
  const uint32_t k_bytes = (uint32_t)min((size_t)T * krow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rk =
      __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv =
      __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdoc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(doc + (size_t)b * T), 0, (uint32_t)T * 4, 0x00020000);
  // lane L of a piece writes LDS chunk L = (row L >> 2, physical chunk L & 3) of a 16-row x 64-byte panel slab
  const int rr = lane >> 2;
  const int lane_chunk = 8 * ((lane & 3) ^ ((rr >> 2) & 3));
  constexpr uint32_t OOB = 0x80000000u;       // >= num_records: the load returns 0 and touches no memory
  // The 128 Q rows as two 64-row panel images in slot 1 (free until the first trip hands it to a K / V tile): 64-byte
  // runs per row instead of 16 bytes per lane at a row stride.  Rows past the segment / the sequence are zero-filled by
  // the descriptor's bounds check.
  {
    const int qrows = min(qleft, T - q0);
    const uint32_t voffq = (uint32_t)(((size_t)rr * qrow_elems + lane_chunk) * 2);
#pragma unroll
    for (int img = 0; img < 2; ++img) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int pc = wave + 4 * i, panel = pc % Tile::NP, rh = pc / Tile::NP;
        const int row = 64 * img + 16 * rh;
        const uint32_t vo = (row + rr < qrows) ? voffq : OOB;
        const uint32_t so = (uint32_t)((((size_t)lq0 + row) * qrow_elems + (size_t)h * D + 32 * panel) * 2);
        char* dst = smem + STAGEB + img * IMGB + panel * (Tile::PSTRIDE * 2) + rh * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      }
    }
  }

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = q0 / kTile, t1 = min(t0 + BM / kTile - 1, meta.nt - 1);
  const int n_pre = __builtin_amdgcn_readfirstlane(kl_head.x);
  const bool pre = n_pre <= kListPre;                    // the stored list is complete: use it
  stamp();                                                   // 1: round trip A is back
  int bminpos = 0x7fffffff, bmax = 0, j_lo = meta.nt, j_hi = t1;
  if (!pre) {
    // ---- long lists, bidirectional masks, key-chunk restrictions: the tile range from the metadata of the two 64-row
    // halves of the query tile, the list built here in chunks of CAP (as attn_fwd.hip)
    for (int t = t0; t <= t1; ++t) {
      bminpos = min(bminpos, m_minpos[t]);
      bmax = max(bmax, m_max[t]);
      j_lo = min(j_lo, meta.q_lo[(size_t)b * meta.nt + t]);
    }
    if (bidir)
      for (int t = t0; t <= t1; ++t) j_hi = max(j_hi, meta.kv_hi[(size_t)b * meta.nt + t]);
  }
  const int qcap = bidir ? 0x7fffffff : qrow;          // `kv <= qcap`: the causal term of the predicate
  stamp();                                                   // 2: tile range known

  // tiles of [lo, hi] that may interact with this query tile -> list entries {tile, min id, max id, min positive id}
  auto build_list = [&](int lo, int hi_t) {
    const int j = lo + tid;
    int mn = 0, mx = 0, mp = 0;
    bool ok = false;
    if (j <= hi_t) {
      mn = m_min[j];
      mx = m_max[j];
      mp = m_minpos[j];
      ok = tile_may_interact(bminpos, bmax, mp, mx) && qv.kv_tile_on(j);
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wcount[w];
      before += w < wave ? c : 0;
      total += c;
    }
    if (ok) tlist[before + __popcll(bal & ((1ull << lane) - 1ull))] = i32x4_t{j, mn, mx, mp};
    const int n = __builtin_amdgcn_readfirstlane(total);
    if (tid < 4) tlist[n + tid] = i32x4_t{j_hi + 1, 0, 0, 0};
    __syncthreads();
    return n;
  };

  const uint32_t voff = (uint32_t)(((size_t)rr * krow_elems + lane_chunk) * 2);
  auto issue = [&](int j, int slot) {           // always IPS instructions (the counted wait below relies on it)
    char* st = smem + slot * STAGEB;
    const int k0 = j * BN;
    const int left = min(T - k0, BN);
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

  const PRowReader<BN, D> krd(l31, hi);
  const PTrReader<BN, D> vrd(lane);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t vtr0 = lds0 + IMGB + 2 * (uint32_t)vrd.t[0], vtr1 = lds0 + IMGB + 2 * (uint32_t)vrd.t[1];
  const uint32_t krw0 = lds0 + 2 * (uint32_t)krd.a[0], krw1 = lds0 + 2 * (uint32_t)krd.a[1];
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  int cur = 0;                 // ring slot of the stage being computed
  bool inflight = false;       // the first stage of the coming run is already travelling
  int n = 0;
  if (pre) {
    // ---- round trip B: the first K / V tile leaves as soon as its index is known; the list goes to LDS under it
    n = n_pre;
    j_lo = j_hi;               // (one pass of the chunk loop below)
    if (n > 0) {
      issue(__builtin_amdgcn_readfirstlane(kl_first.x), 0);
      inflight = true;
    }
    if (tid < n) tlist[tid] = kl_mine;
    if (tid < 4) tlist[n + tid] = i32x4_t{j_hi + 1, 0, 0, 0};
  }
  // wave-level id range of the 32 query rows, in SGPRs — read behind the first K / V issue, which must not wait for it
  // (rows past the segment end count as pad rows)
  const int4 qsc = scalarize(qs4);
  const int wminpos = qsc.x, wmax = qsc.y;
  const bool w_has_zero = qsc.z != 0 || 32 * wave + 32 > qleft || wq0 + 32 > T;
  // the Q rows have landed (the K / V tile issued behind them may stay in flight) -> MFMA B operands in registers
  if (inflight) wait_vmcnt<IPS>(); else wait_vmcnt<0>();
  __syncthreads();
  bf16x8_t qreg[KSTEPS];
  {
    const bf16_t* qimg = reinterpret_cast<const bf16_t*>(smem + STAGEB + (wave >> 1) * IMGB);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qreg[s] = krd.operand(qimg, 32 * (wave & 1), s);
    // (slot 1 is handed to a K / V tile behind the first trip's barrier: the reads must be back before this wave
    // arrives there)
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qreg[s]));
  }
  stamp();                                                   // 3: Q in registers, list in LDS

  f32x16_t oacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  for (int c_lo = j_lo; c_lo <= j_hi; c_lo += CAP) {
    if (!pre) n = build_list(c_lo, min(c_lo + CAP - 1, j_hi));
    if (n == 0) continue;
    int4 e_cur = scalarize(tlist[0]);
    if (!inflight) issue(e_cur.x, cur);
    int4 e_nxt = scalarize(tlist[1]);
    for (int i = 0; i < n; ++i) {
      const i32x4_t e_nn = tlist[i + 2];                  // (vector read now, scalarised at the hand-over)
      // my pieces of this stage have landed ...
      wait_vmcnt<0>();
      stamp();                                             // trip: my pieces have landed
      // ... everybody's have, and everybody has left the previous stage: its slot takes the next one
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stamp();                                             // trip: through the barrier
      // K "row" operands: inline-asm reads in batches of four fragments (k-steps 2 g, 2 g + 1 of both 32-row blocks), two
      // batches in flight — left to hipcc, every MFMA of the chain waited for its own read (read, lgkmcnt(0), MFMA:
      // sixteen exposed LDS latencies per tile).  The first two batches leave before the DMA of the next stage is
      // issued and land under its address arithmetic.
      const int j = e_cur.x, kmin = e_cur.y, kmax = e_cur.z, kminpos = e_cur.w;
      const int k0 = j * BN;
      const bool active = uniform((bidir || k0 <= wq0 + 31) && tile_may_interact(wminpos, wmax, kminpos, kmax));
      const uint32_t ka0 = krw0 + (uint32_t)(cur * STAGEB), ka1 = krw1 + (uint32_t)(cur * STAGEB);
      constexpr bool KBATCH = D == 128;
      u32x4_t kf[2][4];
      auto k_issue = [&](auto GG, auto BUF) {
        constexpr int g = decltype(GG)::value, buf = decltype(BUF)::value;
        static_for<4>([&](auto F) {
          constexpr int f = decltype(F)::value, s2 = 2 * g + (f >> 1), blk = f & 1;
          constexpr int off = 2 * ((s2 >> 1) * Tile::PSTRIDE + 32 * blk * 32);
          kf[buf][f] = ds_b128<off>((s2 & 1) ? ka1 : ka0);
        });
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      if constexpr (KBATCH) {
        if (active) {
          k_issue(I0{}, I0{});
          k_issue(I1{}, I1{});
        }
      }
      inflight = i + 1 < n;
      if (inflight) issue(e_nxt.x, cur ^ 1);

      const int* docs = reinterpret_cast<const int*>(smem + cur * STAGEB + 2 * IMGB + 256 * wave);
      const uint32_t va0 = vtr0 + (uint32_t)(cur * STAGEB), va1 = vtr1 + (uint32_t)(cur * STAGEB);
      if (active) {
        const bool need_mask = uniform(
            !(kmin == kmax && kmax == wminpos && wminpos == wmax && !w_has_zero && (bidir || k0 + BN - 1 <= wq0)));
        // ---- S^T[kv, q] = K[kv, :] . Q[q, :]   (two chains, one per 32-row block, interleaved)
        f32x16_t sacc[2];
#define TN_K_RETIRE(buf, keep)                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(kf[buf][0]), "+v"(kf[buf][1]), "+v"(kf[buf][2]), "+v"(kf[buf][3])       \
               : "n"(keep))
        auto k_mfma = [&](auto GG, auto BUF) {
          constexpr int g = decltype(GG)::value, buf = decltype(BUF)::value;
          static_for<4>([&](auto F) {
            constexpr int f = decltype(F)::value, s2 = 2 * g + (f >> 1), blk = f & 1;
            const bf16x8_t a = __builtin_bit_cast(bf16x8_t, kf[buf][f]);
            if constexpr (s2 == 0) sacc[blk] = mfma32(a, qreg[0], zero16);
            else sacc[blk] = mfma32(a, qreg[s2], sacc[blk]);
          });
        };
        if constexpr (!KBATCH) {
          const bf16_t* Ks = reinterpret_cast<const bf16_t*>(smem + cur * STAGEB);
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            sacc[blk] = mfma32(krd.operand(Ks, 32 * blk, 0), qreg[0], zero16);
#pragma unroll
            for (int s2 = 1; s2 < KSTEPS; ++s2) sacc[blk] = mfma32(krd.operand(Ks, 32 * blk, s2), qreg[s2], sacc[blk]);
          }
        } else {
          TN_K_RETIRE(0, 4);
          k_mfma(I0{}, I0{});
          k_issue(I2{}, I0{});
          TN_K_RETIRE(1, 4);
          k_mfma(I1{}, I1{});
          k_issue(I3{}, I1{});
          TN_K_RETIRE(0, 4);
          k_mfma(I2{}, I0{});
          TN_K_RETIRE(1, 0);
          k_mfma(I3{}, I1{});
        }
#undef TN_K_RETIRE
        // ---- mask, online softmax (lane-local: this lane's query column); scores stay RAW in the accumulator,
        // the softmax scale rides in the exponent's fma: p = exp2(s * c - m), m tracked in the scaled domain
        if (need_mask) {
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const i32x4_t dk = *reinterpret_cast<const i32x4_t*>(docs + 32 * blk + 8 * r4 + 4 * hi);
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
        float mx;
        {   // four independent v_max3 chains
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
        mx = half_max(mx) * scale_log2;
        // Deferred rescale (threshold 8 in the log2 domain), as attn_fwd.hip
        if (uniform(!__all(mx - m_run <= 8.f))) {
          const float m_new = fmaxf(m_run, mx);
          const float alpha = fast_exp2(m_run - m_new);
          m_run = m_new;
          l_run *= alpha;
#pragma unroll
          for (int i2 = 0; i2 < DBLK; ++i2)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[i2][r] *= alpha;
        }
        // p = exp2(s c - m) and the row sum on PAIRS of scores (v_pk_fma_f32 / v_pk_add_f32: half the VALU issues of
        // the scalar forms; a packed op is two IEEE ops)
        const f32x2_t c2 = {scale_log2, scale_log2}, nm2 = {-m_run, -m_run};
        f32x2_t ps2 = {0.f, 0.f};
        bf16x8_t pb[4];                 // pb[2 blk + sp]: kv rows 32 blk + 16 sp .. + 15 (contraction slots of P^T)
#pragma unroll
        for (int bs = 0; bs < 4; ++bs) {
          float p[8];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2_t s2v = {sacc[bs >> 1][8 * (bs & 1) + e], sacc[bs >> 1][8 * (bs & 1) + e + 1]};
            const f32x2_t t2 = __builtin_elementwise_fma(s2v, c2, nm2);
            p[e] = fast_exp2(t2.x);
            p[e + 1] = fast_exp2(t2.y);
            ps2 += f32x2_t{p[e], p[e + 1]};
          }
          const u32x4_t t = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
          pb[bs] = __builtin_bit_cast(bf16x8_t, t);
        }
        l_run += ps2.x + ps2.y;
        // ---- O^T[d, q] += V^T[d, kv] P^T[kv, q]: per 16 kv rows (bs) DBLK independent accumulators; the transpose
        // reads of step bs + 1 are in flight while the MFMAs of step bs run
        __builtin_amdgcn_sched_barrier(0);
        u32x2_t vt[2][DBLK][2];
        auto vt_issue = [&](auto BS, auto BUF) {        // V^T operands of kv rows 16 bs .. 16 bs + 15
          constexpr int bs = decltype(BS)::value, buf = decltype(BUF)::value;
          static_for<DBLK>([&](auto DB) {
            constexpr int db = decltype(DB)::value;
            constexpr int off = 2 * (db * Tile::PSTRIDE + 16 * bs * 32);
            vt[buf][db][0] = ds_tr16<off>(va0);
            vt[buf][db][1] = ds_tr16<off + 512>(va1);
          });
        };
        // ... have arrived (KEEP younger LDS reads may stay in flight).  A macro: an asm operand cannot name a captured array
#define TN_VT_RETIRE(buf, keep)                                                                                          \
  if constexpr (DBLK == 4)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                 \
                 : "+v"(vt[buf][0][0]), "+v"(vt[buf][0][1]), "+v"(vt[buf][1][0]), "+v"(vt[buf][1][1]),                   \
                   "+v"(vt[buf][2 % DBLK][0]), "+v"(vt[buf][2 % DBLK][1]), "+v"(vt[buf][3 % DBLK][0]),                   \
                   "+v"(vt[buf][3 % DBLK][1])                                                                            \
                 : "n"(keep));                                                                                           \
  else                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(%4)"                                                                                 \
                 : "+v"(vt[buf][0][0]), "+v"(vt[buf][0][1]), "+v"(vt[buf][1][0]), "+v"(vt[buf][1][1])                   \
                 : "n"(keep))
        auto vt_mfma = [&](auto BS, auto BUF) {
          constexpr int bs = decltype(BS)::value, buf = decltype(BUF)::value;
          static_for<DBLK>([&](auto DB) {
            constexpr int db = decltype(DB)::value;
            const u32x4_t a4 = {vt[buf][db][0].x, vt[buf][db][0].y, vt[buf][db][1].x, vt[buf][db][1].y};
            oacc[db] = mfma32(__builtin_bit_cast(bf16x8_t, a4), pb[bs], oacc[db]);
          });
        };
        vt_issue(I0{}, I0{});
        vt_issue(I1{}, I1{});
        TN_VT_RETIRE(0, 2 * DBLK);
        vt_mfma(I0{}, I0{});
        vt_issue(I2{}, I0{});
        TN_VT_RETIRE(1, 2 * DBLK);
        vt_mfma(I1{}, I1{});
        vt_issue(I3{}, I1{});
        TN_VT_RETIRE(0, 2 * DBLK);
        vt_mfma(I2{}, I0{});
        TN_VT_RETIRE(1, 0);
        vt_mfma(I3{}, I1{});
#undef TN_VT_RETIRE
      }
      stamp();                                             // trip: computed
      // ---- hand-over
      e_cur = e_nxt;
      e_nxt = scalarize(e_nn);
      cur ^= 1;
    }
    // (a further chunk builds a new list: nothing is in flight, the last trip issued nothing)
    if (!pre) __syncthreads();
  }

  // ---- epilogue: both ring slots are quiet behind the last trip's barrier — slot `cur` was read a trip ago — and the
  // 128 rows leave through it: each wave writes its 32 x D block (8-byte runs of the accumulator layout) into a private
  // image and reads it back as whole rows — 16-byte stores, 64 / (D / 8) rows per instruction, instead of 8 bytes per
  // lane at a row stride (store-issue bound: MI355X_MICROARCH.md "attention epilogue store tail")
  const float l_tot = half_sum(l_run);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  {
    char* ob = smem + cur * STAGEB + wave * (32 * OSTR);
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        u32x2_t o2 = {pack2bf(oacc[db][4 * r4 + 0] * inv, oacc[db][4 * r4 + 1] * inv),
                      pack2bf(oacc[db][4 * r4 + 2] * inv, oacc[db][4 * r4 + 3] * inv)};
        *reinterpret_cast<u32x2_t*>(ob + l31 * OSTR + (32 * db + 8 * r4 + 4 * hi) * 2) = o2;
      }
    constexpr int CPR = D / 8, RPI = 64 / CPR;           // 16-byte chunks per row, rows per store instruction
    const int cc = lane % CPR, r0 = lane / CPR;
    bf16_t* op = O + (((size_t)b * qv.rpb + lq0 + 32 * wave) * Nh + h) * D + cc * 8;
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int row = i * RPI + r0;
      const u32x4_t v4 = *reinterpret_cast<const u32x4_t*>(ob + row * OSTR + cc * 16);
      if (32 * wave + row < qleft && wq0 + row < T) *reinterpret_cast<u32x4_t*>(op + (size_t)row * Nh * D) = v4;
    }
    if (qvalid && hi == 0) LSE2[((size_t)b * Nh + h) * qv.rpb + lrow] = l_tot > 0.f ? m_run + log2f(l_tot) : INFINITY;
  }
  stamp();                                         // exit
}

}  // namespace tn

using namespace tn;

static unsigned long long* g_stream_trace = nullptr;      // tn_attn_fwd_stream_trace (development entry point)

int tn_attn_fwd_stream_launch(const void* q, const void* k, const void* v, void* o, float* lse2, const int* doc,
                              AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, int D, float sl2, hipStream_t st) {
  dim3 grid(Nh, qv.tiles(0, 128) + qv.tiles(1, 128), B), block(256);
#define TN_LAUNCH(DD, TT)                                                                                              \
  hipLaunchKernelGGL((attn_fwd_stream_kernel<DD, TT>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,         \
                     (const bf16_t*)v, (bf16_t*)o, lse2, doc, m, qv, T, Nh, Nkv, sl2, g_stream_trace)
  if (D == 128 && g_stream_trace) TN_LAUNCH(128, true);
  else if (D == 128) TN_LAUNCH(128, false);
  else if (D == 64) TN_LAUNCH(64, false);
  else return TN_EINVAL;
#undef TN_LAUNCH
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// Development entry point (NOT part of the C ABI): D = 128 launches of the stream forward stamp s_memtime per wave into
// `buf` (uint64 [workgroups * 4 * 64]) until it is reset with nullptr.
extern "C" int tn_attn_fwd_stream_trace(void* buf) {
  g_stream_trace = (unsigned long long*)buf;
  return TN_OK;
}
