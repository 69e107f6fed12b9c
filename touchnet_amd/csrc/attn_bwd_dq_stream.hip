// Packed (document-masked) flash attention BACKWARD, dQ pass, for gfx950 — the treatment attn_fwd_stream.hip gave the
// forward, applied to attn_bwd.hip's dQ kernel (same maths, same tile walk, same outputs):
//   * the KV tiles a query tile meets and the id statistics of each wave's rows come from the mask metadata
//     (attn_common.h klist / qstat) instead of an LDS window + scan inside every workgroup;
//   * each wave fetches ITS 32 Q rows and 32 dO rows by LDS-DMA (64-byte runs per row) into a private piece of the ring
//     area and reads them into the MFMA B-operand registers — no 16-bytes-per-lane loads at a row stride, no barrier (a
//     wave reads only what it fetched); the O rows of the delta = rowsum(dO o O) by-product come straight from memory
//     under that round trip;
//   * K / V "row" operands (S^T = K Q^T, dP^T = V dO^T) and K^T operands (dQ^T += K^T dS^T) are inline-asm reads in
//     batches with counted lgkmcnt waits (left to hipcc every MFMA of a chain waited for its own read);
//   * the dQ rows leave through the free ring slot as whole rows, 16 bytes per lane.
// Reference semantics: the dQ loop of flex_attention's backward (transformers/integrations/flex_attention.py:264-340 driven by
// touchnet/models/kimi_audio/modeling_kimi_audio.py:582-585; plain causal for touchnet/models/qwen2_audio/__init__.py:190-193).
#include <stdlib.h>

#include "attn_stream.h"

namespace tn {

template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_stream_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, float* __restrict__ Delta,
    bf16_t* __restrict__ dQ, const int* __restrict__ doc, AttnMeta meta, QView qv, int T, int Nh, int Nkv,
    float scale, float scale_log2, const bf16_t* __restrict__ O, const bf16_t* __restrict__ rcos,
    const bf16_t* __restrict__ rsin) {
  // O != null: this kernel ALSO forms delta = rowsum(dO o O) of its 128 query rows and writes it to `Delta` for the
  // dK / dV pass, which is launched BEHIND it.  O == null: `Delta` is read.
  using namespace fstream;
  constexpr int BM = 128, BN = 64, NST = 2;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  using Tile = PTile<BN, D>;
  using QTile = PTile<32, D>;                   // a wave's private 32-row image of Q / dO
  constexpr int IMGB = Tile::SIZE * 2;          // bytes of one panel image
  constexpr int QIMGB = QTile::SIZE * 2;
  constexpr int NPC = Tile::NP * (BN / 16);     // 1-KiB DMA pieces per image: 16 rows of one panel each
  constexpr int PPW = NPC / 4;                  // pieces per wave and image
  constexpr int OSTR = 2 * D + 16;              // row stride (bytes) of the dQ staging image
  constexpr int STAGE_KV = 2 * IMGB + 4 * 256;  // {K image | V image | doc ids[64] per wave}
  constexpr int STAGEB = STAGE_KV > BM * OSTR ? STAGE_KV : BM * OSTR;
  constexpr int CAP = 192;                      // tile-list chunk
  static_assert(4 * 2 * QIMGB <= NST * STAGEB, "the waves' private Q / dO images live in the ring area");
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
  int lq0, q0, qleft;
  qv.tile(gridDim.y - 1 - blockIdx.y, BM, lq0, q0, qleft);
  const int wq0 = q0 + 32 * wave;          // global position of the wave's first query row
  const int qrow = wq0 + l31;              // global position
  const int lrow = lq0 + 32 * wave + l31;  // row in the local Q / dO / dQ / LSE / delta buffers
  const bool qvalid = (32 * wave + l31 < qleft) && (qrow < T);

  // ---- round trip A: tile list, id statistics, the lane's own id / LSE / delta inputs, the wave's Q and dO rows
  const bool pre_ok = !bidir;                               // (the stored lists are causal)
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
  const int dq = qvalid ? doc[(size_t)b * T + qrow] : 0;
  const float lse2 = qvalid ? LSE2[((size_t)b * Nh + h) * qv.rpb + lrow] : INFINITY;
  float delta_in = 0.f;
  if (O == nullptr && qvalid) delta_in = Delta[((size_t)b * Nh + h) * qv.rpb + lrow];
  uint4 orow[KSTEPS];                       // the O row's slots 8 hi .. 8 hi + 7 of every k-step (delta by-product)
  if (O != nullptr) {
    const size_t off = (((size_t)b * qv.rpb + (qvalid ? lrow : 0)) * Nh + h) * D + 8 * hi;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      orow[s] = make_uint4(0, 0, 0, 0);
      if (qvalid) orow[s] = *reinterpret_cast<const uint4*>(O + off + 16 * s);
    }
  }

  const size_t qrow_elems = (size_t)Nh * D;
  const uint32_t q_bytes = (uint32_t)min((size_t)qv.rpb * qrow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)(dO + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const size_t krow_elems = (size_t)Nkv * D;
  const uint32_t k_bytes = (uint32_t)min((size_t)T * krow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rk =
      __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv =
      __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdoc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(doc + (size_t)b * T), 0, (uint32_t)T * 4, 0x00020000);
  const int rr = lane >> 2;
  const int lane_chunk = 8 * ((lane & 3) ^ ((rr >> 2) & 3));
  constexpr uint32_t OOB = 0x80000000u;
  char* qpriv = smem + wave * (2 * QIMGB);      // {Q image | dO image} of this wave's 32 rows
  {
    const int wrows = min(qleft - 32 * wave, T - wq0);      // valid rows of this wave (<= 0: none)
    const uint32_t voffq = (uint32_t)(((size_t)rr * qrow_elems + lane_chunk) * 2);
#pragma unroll
    for (int pc = 0; pc < QTile::NP * 2; ++pc) {
      const int panel = pc % QTile::NP, rh = pc / QTile::NP;
      const uint32_t vo = (16 * rh + rr < wrows) ? voffq : OOB;
      const uint32_t so =
          (uint32_t)((((size_t)lq0 + 32 * wave + 16 * rh) * qrow_elems + (size_t)h * D + 32 * panel) * 2);
      char* dst = qpriv + panel * (QTile::PSTRIDE * 2) + rh * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdo, (lds_ptr_t)(dst + QIMGB), 16, vo, so, 0, 0);
    }
  }

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int t0 = q0 / kTile, t1 = min(t0 + 1, meta.nt - 1);
  const int n_pre = __builtin_amdgcn_readfirstlane(kl_head.x);
  const bool pre = n_pre <= kListPre;
  int bminpos = 0x7fffffff, bmax = 0, j_lo = meta.nt, j_hi = t1;
  if (!pre) {
    for (int t = t0; t <= t1; ++t) {
      bminpos = min(bminpos, m_minpos[t]);
      bmax = max(bmax, m_max[t]);
      j_lo = min(j_lo, meta.q_lo[(size_t)b * meta.nt + t]);
    }
    if (bidir)
      for (int t = t0; t <= t1; ++t) j_hi = max(j_hi, meta.kv_hi[(size_t)b * meta.nt + t]);
  }
  const int qcap = bidir ? 0x7fffffff : qrow;               // `kv <= qcap`: the causal term of the predicate

  auto build_list = [&](int lo, int hi_t) {
    const int j = lo + tid;
    int mn = 0, mx = 0, mp = 0;
    bool ok = false;
    if (j <= hi_t) {
      mn = m_min[j];
      mx = m_max[j];
      mp = m_minpos[j];
      ok = tile_may_interact(bminpos, bmax, mp, mx);
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
  auto issue = [&](int j, int slot) {
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

  // ---- the wave's Q / dO rows have landed (its own DMA: no barrier) -> MFMA B operands in registers
  wait_vmcnt<0>();
  bf16x8_t qreg[KSTEPS], doreg[KSTEPS];
  {
    const PRowReader<32, D> qrd(l31, hi);
    const bf16_t* qi = reinterpret_cast<const bf16_t*>(qpriv);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      qreg[s] = qrd.operand(qi, 0, s);
      doreg[s] = qrd.operand(qi + QTile::SIZE, 0, s);
    }
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qreg[s]), "+v"(doreg[s]));
  }
  float delta = delta_in;
  if (O != nullptr) {
    float dsum = 0.f;
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
    delta = dsum + __shfl_xor(dsum, 32, 64);            // (both 32-lane halves hold the same rows)
    if (qvalid && hi == 0) Delta[((size_t)b * Nh + h) * qv.rpb + lrow] = delta;
  }
  // everybody has read its private images: the ring area is free for K / V tiles and the list
  __syncthreads();

  int cur = 0;
  bool inflight = false;
  int n = 0;
  if (pre) {
    // ---- round trip B: the first K / V tile; the list goes to LDS under it
    n = n_pre;
    j_lo = j_hi;
    if (n > 0) {
      issue(__builtin_amdgcn_readfirstlane(kl_first.x), 0);
      inflight = true;
    }
    if (tid < n) tlist[tid] = kl_mine;
    if (tid < 4) tlist[n + tid] = i32x4_t{j_hi + 1, 0, 0, 0};
    __syncthreads();
  }
  const int4 qsc = scalarize(qs4);
  const int wminpos = qsc.x, wmax = qsc.y;
  const bool w_has_zero = qsc.z != 0 || 32 * wave + 32 > qleft || wq0 + 32 > T;

  f32x16_t dqacc[DBLK];
#pragma unroll
  for (int i = 0; i < DBLK; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  const PRowReader<BN, D> rrd(l31, hi);
  const PTrReader<BN, D> trd(lane);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t ktr0 = lds0 + 2 * (uint32_t)trd.t[0], ktr1 = lds0 + 2 * (uint32_t)trd.t[1];
  const uint32_t krw0 = lds0 + 2 * (uint32_t)rrd.a[0], krw1 = lds0 + 2 * (uint32_t)rrd.a[1];
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  for (int c_lo = j_lo; c_lo <= j_hi; c_lo += CAP) {
    if (!pre) n = build_list(c_lo, min(c_lo + CAP - 1, j_hi));
    if (n == 0) continue;
    int4 e_cur = scalarize(tlist[0]);
    if (!inflight) issue(e_cur.x, cur);
    int4 e_nxt = scalarize(tlist[1]);
    for (int i = 0; i < n; ++i) {
      const i32x4_t e_nn = tlist[i + 2];
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int j = e_cur.x, kmin = e_cur.y, kmax = e_cur.z, kminpos = e_cur.w;
      const int k0 = j * BN;
      const bool active = uniform((bidir || k0 <= wq0 + 31) && tile_may_interact(wminpos, wmax, kminpos, kmax));
      const uint32_t ka0 = krw0 + (uint32_t)(cur * STAGEB), ka1 = krw1 + (uint32_t)(cur * STAGEB);
      const uint32_t ta0 = ktr0 + (uint32_t)(cur * STAGEB), ta1 = ktr1 + (uint32_t)(cur * STAGEB);
      const bool act0 = active && uniform(bidir || k0 <= wq0 + 31);
      const bool act1 = active && uniform(bidir || k0 + 32 <= wq0 + 31);
      inflight = i + 1 < n;
      if (inflight) issue(e_nxt.x, cur ^ 1);

      const int* docs = reinterpret_cast<const int*>(smem + cur * STAGEB + 2 * IMGB + 256 * wave);
      if (active) {
        const bool need_mask = uniform(!(kmin == kmax && kmax == wminpos && wminpos == wmax && !w_has_zero &&
                                         (bidir || k0 + BN - 1 <= wq0)));
#define TN_K_RETIRE(buf, keep)                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(kf[buf][0]), "+v"(kf[buf][1]), "+v"(kf[buf][2]), "+v"(kf[buf][3])       \
               : "n"(keep))
        auto block = [&](auto BLK, bool act) {
          constexpr int blk = decltype(BLK)::value;
          if (!act) return;                                   // this 32-row KV block is above the diagonal
          // row operands of the block in batches of four fragments: {K, V} x k-steps {2 g, 2 g + 1}, two batches in flight
          // (declared HERE: an asm operand cannot name an array captured from an enclosing scope)
          u32x4_t kf[2][4];
          auto k_issue = [&](auto GG, auto BUF) {
            constexpr int g = decltype(GG)::value, buf = decltype(BUF)::value;
            static_for<4>([&](auto F) {
              constexpr int f = decltype(F)::value, s2 = 2 * g + (f >> 1), img = f & 1;
              constexpr int off = 2 * ((s2 >> 1) * Tile::PSTRIDE + 32 * blk * 32) + img * IMGB;
              kf[buf][f] = ds_b128<off>((s2 & 1) ? ka1 : ka0);
            });
          };
          f32x16_t sacc, dpacc;
          auto k_mfma = [&](auto GG, auto BUF) {
            constexpr int g = decltype(GG)::value, buf = decltype(BUF)::value;
            static_for<2>([&](auto SS) {
              constexpr int ss = decltype(SS)::value, s2 = 2 * g + ss;
              const bf16x8_t ak = __builtin_bit_cast(bf16x8_t, kf[buf][2 * ss]);
              const bf16x8_t av = __builtin_bit_cast(bf16x8_t, kf[buf][2 * ss + 1]);
              if constexpr (s2 == 0) {
                sacc = mfma32(ak, qreg[0], zero16);
                dpacc = mfma32(av, doreg[0], zero16);
              } else {
                sacc = mfma32(ak, qreg[s2], sacc);
                dpacc = mfma32(av, doreg[s2], dpacc);
              }
            });
          };
          using I2 = std::integral_constant<int, 2>;
          using I3 = std::integral_constant<int, 3>;
          k_issue(I0{}, I0{});
          k_issue(I1{}, I1{});
          if constexpr (KSTEPS == 8) {
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
          } else {
            TN_K_RETIRE(0, 4);
            k_mfma(I0{}, I0{});
            TN_K_RETIRE(1, 0);
            k_mfma(I1{}, I1{});
          }
          // K^T operands of the first 16 kv rows of the block travel under the dS arithmetic
          u32x2_t kt[2][DBLK][2];
          auto kt_issue = [&](auto SP, auto BUF) {
            constexpr int sp = decltype(SP)::value, buf = decltype(BUF)::value;
            static_for<DBLK>([&](auto DB) {
              constexpr int db = decltype(DB)::value;
              constexpr int off = 2 * (db * Tile::PSTRIDE + (32 * blk + 16 * sp) * 32);
              kt[buf][db][0] = ds_tr16<off>(ta0);
              kt[buf][db][1] = ds_tr16<off + 512>(ta1);
            });
          };
          kt_issue(I0{}, I0{});
          kt_issue(I1{}, I1{});
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
          bf16x8_t dsb[2];
#pragma unroll
          for (int sp = 0; sp < 2; ++sp) {
            const u32x4_t u = {pack2bf(ds[8 * sp + 0], ds[8 * sp + 1]), pack2bf(ds[8 * sp + 2], ds[8 * sp + 3]),
                               pack2bf(ds[8 * sp + 4], ds[8 * sp + 5]), pack2bf(ds[8 * sp + 6], ds[8 * sp + 7])};
            dsb[sp] = __builtin_bit_cast(bf16x8_t, u);
          }
          __builtin_amdgcn_sched_barrier(0);
#define TN_KT_RETIRE(buf, keep)                                                                                          \
  if constexpr (DBLK == 4)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(%8)"                                                                                 \
                 : "+v"(kt[buf][0][0]), "+v"(kt[buf][0][1]), "+v"(kt[buf][1][0]), "+v"(kt[buf][1][1]),                   \
                   "+v"(kt[buf][2 % DBLK][0]), "+v"(kt[buf][2 % DBLK][1]), "+v"(kt[buf][3 % DBLK][0]),                   \
                   "+v"(kt[buf][3 % DBLK][1])                                                                            \
                 : "n"(keep));                                                                                           \
  else                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(%4)"                                                                                 \
                 : "+v"(kt[buf][0][0]), "+v"(kt[buf][0][1]), "+v"(kt[buf][1][0]), "+v"(kt[buf][1][1])                   \
                 : "n"(keep))
          // dQ^T[d, q] += K^T[d, kv] dS^T[kv, q]
          TN_KT_RETIRE(0, 2 * DBLK);
          static_for<DBLK>([&](auto DB) {
            constexpr int db = decltype(DB)::value;
            const u32x4_t a4 = {kt[0][db][0].x, kt[0][db][0].y, kt[0][db][1].x, kt[0][db][1].y};
            dqacc[db] = mfma32(__builtin_bit_cast(bf16x8_t, a4), dsb[0], dqacc[db]);
          });
          TN_KT_RETIRE(1, 0);
          static_for<DBLK>([&](auto DB) {
            constexpr int db = decltype(DB)::value;
            const u32x4_t a4 = {kt[1][db][0].x, kt[1][db][0].y, kt[1][db][1].x, kt[1][db][1].y};
            dqacc[db] = mfma32(__builtin_bit_cast(bf16x8_t, a4), dsb[1], dqacc[db]);
          });
#undef TN_KT_RETIRE
        };
        block(I0{}, act0);
        block(I1{}, act1);
#undef TN_K_RETIRE
      }
      e_cur = e_nxt;
      e_nxt = scalarize(e_nn);
      cur ^= 1;
    }
    if (!pre) __syncthreads();
  }

  // ---- epilogue: the dQ rows leave through the free ring slot as whole rows (attn_fwd_stream.hip)
  {
    constexpr int CPR = D / 8, RPI = 64 / CPR, NI = 32 / RPI;
    const int cc = lane % CPR, r0 = lane / CPR;
    // rcos / rsin (tn_attn_bwd_rope, D = 128): the rows leave as the gradient of the UN-rotated q.  The table entries of
    // this lane's eight chunks are asked for first, all at once, and arrive while the accumulators are staged (fetched
    // chunk by chunk inside the store loop they cost the dK / dV kernel 50 us of a 476 us launch, one L2 round trip each)
    const bool rot = D == 128 && rcos != nullptr;
    u32x4_t c4[NI] = {}, s4[NI] = {};
    if (rot) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const size_t to = ((size_t)b * T + min(wq0 + i * RPI + r0, T - 1)) * (D / 2) + (cc & (CPR / 2 - 1)) * 8;
        c4[i] = *reinterpret_cast<const u32x4_t*>(rcos + to);
        s4[i] = *reinterpret_cast<const u32x4_t*>(rsin + to);
      }
    }
    char* ob = smem + cur * STAGEB + wave * (32 * OSTR);
#pragma unroll
    for (int db = 0; db < DBLK; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        u32x2_t o2 = {pack2bf(dqacc[db][4 * r4 + 0] * scale, dqacc[db][4 * r4 + 1] * scale),
                      pack2bf(dqacc[db][4 * r4 + 2] * scale, dqacc[db][4 * r4 + 3] * scale)};
        *reinterpret_cast<u32x2_t*>(ob + l31 * OSTR + (32 * db + 8 * r4 + 4 * hi) * 2) = o2;
      }
    bf16_t* op = dQ + (((size_t)b * qv.rpb + lq0 + 32 * wave) * Nh + h) * D + cc * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = i * RPI + r0;
      u32x4_t v4 = *reinterpret_cast<const u32x4_t*>(ob + row * OSTR + cc * 16);
      if (rot) {
        // tn_rope_apply(backward)'s arithmetic on the staged bf16 values, the other half of the row from the same image:
        // the bits the row kernel would produce
        const u32x4_t p4 = *reinterpret_cast<const u32x4_t*>(ob + row * OSTR + (cc ^ (CPR / 2)) * 16);
        v4 = rope_grad_chunk(v4, p4, c4[i], s4[i], cc >= CPR / 2);
      }
      if (32 * wave + row < qleft && wq0 + row < T) *reinterpret_cast<u32x4_t*>(op + (size_t)row * Nh * D) = v4;
    }
  }
}

void launch_attn_bwd_dq_stream(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO, const float* lse2,
                               float* delta, bf16_t* dQ, const int* doc, AttnMeta m, QView qv, int B, int T, int Nh,
                               int Nkv, int D, float scale, float sl2, const bf16_t* O, const bf16_t* rcos, const bf16_t* rsin,
                               hipStream_t st) {
  dim3 gq(Nh, qv.tiles(0, 128) + qv.tiles(1, 128), B), block(256);
  if (D == 128)
    hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<128>), gq, block, 0, st, Q, K, V, dO, lse2, delta, dQ, doc, m, qv, T,
                       Nh, Nkv, scale, sl2, O, rcos, rsin);
  else
    hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<64>), gq, block, 0, st, Q, K, V, dO, lse2, delta, dQ, doc, m, qv, T,
                       Nh, Nkv, scale, sl2, O, nullptr, nullptr);
}

}  // namespace tn
