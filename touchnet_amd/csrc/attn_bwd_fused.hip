// Packed (document-masked, causal) flash attention BACKWARD for gfx950, D = 128: ONE pass for dK and dV.
//
// The two-launch dV / dK scheme of attn_bwd.hip recomputes S = Q K^T in both launches: 16 + 24 MFMAs per
// (32 query x 32 key) block pair.  This kernel computes S and dP once and accumulates both gradients: 32 MFMAs, one
// softmax recomputation instead of two.  What made that impossible inside 256 registers (K and V operands 64, dK^T and
// dV^T accumulators 128, two score tiles in flight) is solved the way the four-wave GEMM experiment showed works under
// hipcc (csrc/gemm_variants): ONE wave per SIMD with the 512-register file,
//   * the 128 accumulators are the LITERAL registers a[0:127] (dV^T blocks a[16 db ..], dK^T blocks a[64 + 16 db ..]),
//     touched only by inline-asm MFMAs that list them as clobbers: hipcc can neither move nor spill them, and its own
//     code stays below 256 VGPRs so it never needs the accumulator file (audited in tests/test_isa_checks.py);
//   * every MFMA is inline asm (with asm AGPR users in the kernel hipcc selects the AGPR form for its own MFMA builtins
//     and would allocate a[0:15] for them); S / dP results land in VGPRs ("=v");
//   * with one wave per SIMD nothing hides the softmax behind another wave's MFMAs, so the loop is software-pipelined
//     by hand over 32-row query stages: while the matrix pipe runs {dV(c), S(n), dK(c), dP(n)} for the current stage c
//     and the next stage n, the gaps between the MFMAs carry dS(c) = P o (dP - delta), P(n) = exp2(S c - LSE), the LDS
//     operand reads of the following group and the LDS-DMA of the stage three ahead.  `sched_barrier(0)` after every
//     gap pins the order (asm statements are invisible to sched_group_barrier).
// Stream of stages, LDS-DMA ring (four 17.5 KB slots: c, n and two in flight), stage list, tile images, document-mask
// rules and the output contract are those of attn_bwd.hip's dK/dV kernel (same reference semantics:
// transformers/integrations/flex_attention.py:190-201 via touchnet/models/kimi_audio/modeling_kimi_audio.py:582-585).
//
// Round 6 (the treatment of attn_fwd_stream.hip; one workgroup per CU here, so nothing hides a workgroup's start and end):
// the query tiles a KV tile meets and the id statistics of the four waves' rows come from the mask metadata
// (attn_common.h qlist / qstat) instead of dependent loads inside the workgroup; each wave fetches ITS 32 K and V rows by
// LDS-DMA into a private piece of the ring area (64-byte runs instead of 16 bytes per lane at a row stride); dK / dV leave
// through the ring area as whole rows, 16 bytes per lane.
//
// Hazards hipcc does not see (cdna_hip_programming.md 5.7): an asm MFMA's VGPR result is read by compiler VALU code only
// behind >= 8 further MFMAs or an explicit s_nop pad tied to the result; packed P / dS operands are written >= one gap
// before the MFMA that reads them and the first MFMA of a group opens with s_nop 1; transpose reads are asm loads
// retired by one lgkmcnt(0) statement naming every destination.
#include <utility>

#include "attn_common.h"

// Timing experiments only (scripts/r04_fkv_ablate.sh builds variants with -DTN_FKV_ABL=<bits>; results are wrong):
// 1 no LDS-DMA in the trips, 2 no barrier, 4 no softmax arithmetic, 8 no LDS reads, 16 no MFMAs, 32 no lgkmcnt waits
#ifndef TN_FKV_ABL
#define TN_FKV_ABL 0
#endif

namespace tn {

namespace fusedkv {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#define TN_FKV_BLOCKS(M)                                                                                              \
  if constexpr (BLK == 0) { M("a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15"); } \
  if constexpr (BLK == 1) { M("a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31"); } \
  if constexpr (BLK == 2) { M("a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47"); } \
  if constexpr (BLK == 3) { M("a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"); } \
  if constexpr (BLK == 4) { M("a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79"); } \
  if constexpr (BLK == 5) { M("a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95"); } \
  if constexpr (BLK == 6) { M("a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111"); } \
  if constexpr (BLK == 7) { M("a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"); }

// a[16 BLK .. 16 BLK + 15] += A B   (FIRST: opens with the pad a freshly written VGPR operand needs)
template <int BLK, bool FIRST>
__device__ __forceinline__ void acc_mfma(bf16x8_t a, bf16x8_t b) {
#define TN_M(...)                                                                                                     \
  if constexpr ((TN_FKV_ABL & 16) != 0)                                                                               \
    asm volatile("" ::"v"(a), "v"(b));                                                                                \
  else if constexpr (FIRST)                                                                                           \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b),             \
                 "n"(16 * BLK), "n"(16 * BLK + 15) : __VA_ARGS__);                                                    \
  else                                                                                                                \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(16 * BLK),          \
                 "n"(16 * BLK + 15) : __VA_ARGS__)
  TN_FKV_BLOCKS(TN_M)
#undef TN_M
}
template <int BLK>
__device__ __forceinline__ void acc_zero() {
#define TN_M(...)                                                                                                      \
  asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c0+1], 0\n\tv_accvgpr_write_b32 a[%c0+2], 0\n\t"  \
               "v_accvgpr_write_b32 a[%c0+3], 0\n\tv_accvgpr_write_b32 a[%c0+4], 0\n\tv_accvgpr_write_b32 a[%c0+5], 0\n\t" \
               "v_accvgpr_write_b32 a[%c0+6], 0\n\tv_accvgpr_write_b32 a[%c0+7], 0\n\tv_accvgpr_write_b32 a[%c0+8], 0\n\t" \
               "v_accvgpr_write_b32 a[%c0+9], 0\n\tv_accvgpr_write_b32 a[%c0+10], 0\n\tv_accvgpr_write_b32 a[%c0+11], 0\n\t" \
               "v_accvgpr_write_b32 a[%c0+12], 0\n\tv_accvgpr_write_b32 a[%c0+13], 0\n\tv_accvgpr_write_b32 a[%c0+14], 0\n\t" \
               "v_accvgpr_write_b32 a[%c0+15], 0" ::"n"(16 * BLK) : __VA_ARGS__)
  TN_FKV_BLOCKS(TN_M)
#undef TN_M
}
#undef TN_FKV_BLOCKS
template <int IDX>
__device__ __forceinline__ float acc_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "n"(IDX));
  return v;
}

// MFMAs with a VGPR result.  "=&v": the result tuple may not overlap the operands.
// `a` comes from an asm LDS read: the statement opens with the lgkmcnt wait that retires it (K = LDS operations issued
// behind that read; one statement, so hipcc has no boundary to pad between the wait and the MFMA).
template <int K>
__device__ __forceinline__ void mfma_first(f32x16_t& d, u32x4_t a, bf16x8_t b) {
  constexpr int W = (TN_FKV_ABL & 32) ? 15 : K;
  if constexpr ((TN_FKV_ABL & 16) != 0)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "=&v"(d) : "v"(a), "v"(b), "n"(W));
  else
    asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b), "n"(W));
}
template <int K>
__device__ __forceinline__ void mfma_acc(f32x16_t& d, u32x4_t a, bf16x8_t b) {
  constexpr int W = (TN_FKV_ABL & 32) ? 15 : K;
  if constexpr ((TN_FKV_ABL & 16) != 0)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(d) : "v"(a), "v"(b), "n"(W));
  else
    asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b), "n"(W));
}
// >= 13 wait states between the last MFMA of a chain and the first VALU read of its result
__device__ __forceinline__ void mfma_result_pad(f32x16_t& d) { asm volatile("s_nop 7\n\ts_nop 4" : "+v"(d)); }

template <int OFF>
__device__ __forceinline__ u32x2_t ds_tr16(uint32_t addr) {
  u32x2_t r;
  if constexpr ((TN_FKV_ABL & 8) != 0) asm volatile("" : "=v"(r) : "v"(addr));
  else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <class V, int OFF>
__device__ __forceinline__ V ds_b128(uint32_t addr) {
  V r;
  if constexpr ((TN_FKV_ABL & 8) != 0) asm volatile("" : "=v"(r) : "v"(addr));
  else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// LDS operations issued in gap g of group 2 of a trip with a current AND a next stage (see the table in the kernel)
constexpr int g2_ops(int g, bool mask) { return g < 4 ? 4 : g < 6 ? 3 + (mask ? 1 : 0) : 1 + (mask ? 1 : 0); }
// ... and how many of the trip's LDS operations are younger than Q row operand s when S MFMA s is issued
constexpr int g2_wait(int s, bool mask) {
  int y = 0;
  if (s < 4) {                         // loaded in gap 4 + s of group 1, operands s + 1 .. 3 right behind it
    y = 3 - s;
    for (int g = 0; g < s; ++g) y += g2_ops(g, mask);
  } else {                             // loaded first thing in gap s - 4 of group 2
    y = g2_ops(s - 4, mask) - 1;
    for (int g = s - 3; g < s; ++g) y += g2_ops(g, mask);
  }
  return y;
}
static_assert(g2_wait(0, false) == 3 && g2_wait(3, false) == 12 && g2_wait(4, false) == 15 && g2_wait(7, false) == 10 &&
              g2_wait(5, true) == 15 && g2_wait(6, true) == 15 && g2_wait(7, true) == 13, "LDS wait table");

// One stage of the stream as the list holds it (16 bytes) — every field is wave-uniform once read (SGPRs):
//   qsb   global position of the stage's first query row
//   meta  bits 0..7: valid rows (1..32; 0 = a stage that loads nothing)
//         bit 8 + w : wave w of the workgroup takes part in the stage
//         bit 12 + w: wave w needs the element-wise mask predicate for it
//   base  byte offset of the stage's first Q / dO row (this head) inside the batch row's slice
//   srow  byte offset of its first LSE / delta element
struct QStage {
  int qsb, meta, base, srow;
};

}  // namespace fusedkv

template <int D>
__global__ __launch_bounds__(256, 1) void attn_bwd_kv_fused_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ LSE2, const float* __restrict__ Delta,
    bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, const int* __restrict__ doc, AttnMeta meta, QView qv, int T,
    int Nh, int Nkv, float scale, float scale_log2, const bf16_t* __restrict__ rcos, const bf16_t* __restrict__ rsin) {
  using namespace fusedkv;
  static_assert(D == 128, "the fused dK/dV pass is built for head_dim 128");
  constexpr int BNK = 128, BQ = 32, NST = 4, SPT = kTile / BQ;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  using Tile = PTile<BQ, D>;
  constexpr int IMGB = Tile::SIZE * 2;          // bytes of one panel image
  constexpr int NPC = Tile::NP * (BQ / 16);     // 1-KiB DMA pieces per image: 16 rows of one panel each
  constexpr int PPW = NPC / 4;                  // pieces per wave and image
  constexpr int IPS = 2 * PPW + 1;              // DMA instructions per wave and stage (Q, dO pieces + one aux row)
  constexpr int STAGEB = 2 * IMGB + 4 * 256;    // {Q image | dO image | lse | delta | doc | doc}, 256-byte aux rows
  constexpr int LCAP = 256;                     // stage-list chunk: one candidate stage per thread
  // ONE LDS variable (attn_bwd.hip explains why two would serialise the DMA ring)
  using KTile = PTile<32, D>;                   // a wave's private 32-row image of K / V (prologue)
  constexpr int KIMGB = KTile::SIZE * 2;
  constexpr int OSTR = 2 * D + 16;              // row stride (bytes) of the dV / dK staging images (epilogue)
  static_assert(4 * 2 * KIMGB <= NST * STAGEB && 4 * 2 * 32 * OSTR <= NST * STAGEB, "private images live in the ring area");
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGEB + (LCAP + NST + 1) * 16 + 16 + kListPre * 16];
  i32x4_t* slist = reinterpret_cast<i32x4_t*>(smem + NST * STAGEB);     // stage list, one QStage per entry
  int* wcount = reinterpret_cast<int*>(smem + NST * STAGEB + (LCAP + NST + 1) * 16);
  i32x4_t* qent = reinterpret_cast<i32x4_t*>(smem + NST * STAGEB + (LCAP + NST + 1) * 16 + 16);   // the stored q-tile list

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int kt = blockIdx.y, hk = blockIdx.x, b = blockIdx.z;   // (head in x, heavy tiles first: attn_common.h)
  const int G = Nh / Nkv;
  const int k0 = kt * BNK;
  const int wk0 = k0 + 32 * wave;
  const int kvrow = wk0 + l31;
  const bool kvalid = kvrow < T;

  // ---- round trip A: the stored list of query tiles, the id statistics of the four waves' rows, this lane's id, and the
  // wave's K / V rows — issued together
  const bool plain = qv.nseg == 1 && qv.off[0] == 0 && qv.row0[0] == 0;     // (this pass is causal)
  i32x4_t ql_head = {kListPre + 1, 0, 0, 0}, ql_mine = {0, 0, 0, 0};
  if (plain) {
    const i32x4_t* ql = reinterpret_cast<const i32x4_t*>(meta.qlist) + ((size_t)b * meta.nq128 + kt) * (1 + kListPre);
    ql_head = ql[0];
    if (tid < kListPre) ql_mine = ql[1 + tid];
  }
  i32x4_t ws4[4];                             // {min positive id, max id, pad present, -} of every wave's 32 kv rows
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    ws4[w] = i32x4_t{0x7fffffff, 0, 1, 0};
    if (k0 + 32 * w < T) ws4[w] = reinterpret_cast<const i32x4_t*>(meta.qstat)[(size_t)b * meta.nq32 + (k0 + 32 * w) / 32];
  }
  const int dkdoc = kvalid ? doc[(size_t)b * T + kvrow] : 0;
  {
    const size_t krow_elems = (size_t)Nkv * D;
    const uint32_t k_bytes = (uint32_t)min((size_t)T * krow_elems * 2, (size_t)0x7fffffff);
    const __amdgpu_buffer_rsrc_t rk =
        __builtin_amdgcn_make_buffer_rsrc((void*)(K + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv =
        __builtin_amdgcn_make_buffer_rsrc((void*)(V + (size_t)b * T * krow_elems), 0, k_bytes, 0x00020000);
    const int rrk = lane >> 2;
    const uint32_t voffk = (uint32_t)(((size_t)rrk * krow_elems + 8 * ((lane & 3) ^ ((rrk >> 2) & 3))) * 2);
    char* kpriv = smem + wave * (2 * KIMGB);    // {K image | V image} of this wave's 32 rows
#pragma unroll
    for (int pc = 0; pc < KTile::NP * 2; ++pc) {
      const int panel = pc % KTile::NP, rh = pc / KTile::NP;
      const uint32_t vo = (wk0 + 16 * rh + rrk < T) ? voffk : 0x80000000u;
      const uint32_t so = (uint32_t)((((size_t)wk0 + 16 * rh) * krow_elems + (size_t)hk * D + 32 * panel) * 2);
      char* dst = kpriv + panel * (KTile::PSTRIDE * 2) + rh * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dst + KIMGB), 16, vo, so, 0, 0);
    }
  }
  wait_vmcnt<0>();
  asm volatile("" ::"v"(dkdoc));               // (hipcc's own wait for this load must sit HERE, not inside the stage loop)
  bf16x8_t kreg[KSTEPS], vreg[KSTEPS];
  {
    const PRowReader<32, D> krd(l31, hi);
    const bf16_t* ki = reinterpret_cast<const bf16_t*>(smem + wave * (2 * KIMGB));
#pragma unroll
    for (int s2 = 0; s2 < KSTEPS; ++s2) {
      kreg[s2] = krd.operand(ki, 0, s2);
      vreg[s2] = krd.operand(ki + KTile::SIZE, 0, s2);
    }
#pragma unroll
    for (int s2 = 0; s2 < KSTEPS; ++s2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kreg[s2]), "+v"(vreg[s2]));
  }
  const int n_pre = __builtin_amdgcn_readfirstlane(ql_head.x);
  const bool pre = n_pre <= kListPre;          // the stored list is complete: use it
  if (pre && tid < n_pre) qent[tid] = ql_mine;
  // everybody has read its private images (the ring area is free) and the list is in LDS
  __syncthreads();

  const int* m_min = meta.tmin + (size_t)b * meta.nt;
  const int* m_max = meta.tmax + (size_t)b * meta.nt;
  const int* m_minpos = meta.tminpos + (size_t)b * meta.nt;
  const int qt_lo = k0 / kTile;                               // first 64-position query tile (q >= kv), global index
  const int t0 = 2 * kt, t1 = min(2 * kt + 1, meta.nt - 1);
  int bminpos = 0, bmax = 0, qt_end = 0;
  if (!pre) {
    bminpos = min(m_minpos[t0], m_minpos[t1]);
    bmax = max(m_max[t0], m_max[t1]);
    const int qhi64 = max(meta.kv_hi[(size_t)b * meta.nt + t0], meta.kv_hi[(size_t)b * meta.nt + t1]);
    qt_end = min(qhi64 + 1, meta.nt);                         // exclusive
  }
  const int4 wsc = scalarize(ws4[0]);
  const int4 wsc1 = scalarize(ws4[1]), wsc2 = scalarize(ws4[2]), wsc3 = scalarize(ws4[3]);
  const int4 wsw[4] = {wsc, wsc1, wsc2, wsc3};
  int seg_lo[2], seg_n[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int first = qv.off[s2] / kTile, cnt = qv.tiles(s2, kTile);
    seg_lo[s2] = max(qt_lo, first);
    seg_n[s2] = max(min(qt_end, first + cnt) - seg_lo[s2], 0);
  }
  const int nqt = pre ? n_pre : seg_n[0] + seg_n[1];
  const int per_head = SPT * nqt, total_c = per_head * G;
  auto build_list = [&](int cb) {       // candidates [cb, cb + LCAP) -> n entries (+ NST + 1 empty ones behind them)
    const int c = cb + tid;
    QStage d = {0, 0, 0, 0};
    bool valid = false;
    if (c < total_c) {
      const int g = c / per_head, r = c - g * per_head;
      const int idx = r / SPT, part = r % SPT;
      int sg = 0, t64, lt, left, mp, mx, mn;
      if (pre) {                                  // (plain launch: local row = global position)
        const i32x4_t qe = qent[idx];
        t64 = qe.x; mn = qe.y; mx = qe.z; mp = qe.w;
        lt = t64;
        left = min(T - t64 * kTile - BQ * part, BQ);
        valid = left > 0;
      } else {
        sg = idx >= seg_n[0] ? 1 : 0;
        t64 = seg_lo[sg] + idx - (sg ? seg_n[0] : 0);
        lt = t64 - qv.off[sg] / kTile;
        left = min(min(qv.rows[sg] - lt * kTile, T - t64 * kTile) - BQ * part, BQ);
        mp = m_minpos[t64]; mx = m_max[t64]; mn = m_min[t64];
        valid = left > 0 && tile_may_interact(mp, mx, bminpos, bmax);
      }
      const int lrow = qv.row0[sg] + lt * kTile + BQ * part, h = hk * G + g;
      d.qsb = t64 * kTile + BQ * part;
      d.base = (int)(((size_t)lrow * Nh + h) * D * 2);
      d.srow = (int)(((size_t)h * qv.rpb + lrow) * 4);
      int flags = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {             // (same rules as attn_bwd.hip's per-wave tests, evaluated once per stage)
        const int4 ws = wsw[w];
        const int w0 = k0 + 32 * w;
        const bool w_uni = ws.x == ws.y && ws.z == 0;             // all 32 kv rows of wave w in one document
        const bool act = d.qsb + BQ - 1 >= w0 && tile_may_interact(mp, mx, ws.x, ws.y);
        const bool q_uniform = w_uni && mn == mx && mx == ws.y && left == BQ;
        const bool msk = !(q_uniform && d.qsb >= w0 + 31);
        flags |= (act ? 1 << (8 + w) : 0) | (msk ? 1 << (12 + w) : 0);
      }
      d.meta = left | flags;
    }
    const int wv = tid >> 6;
    const unsigned long long bal = __ballot(valid);
    if (lane == 0) wcount[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int cnt = wcount[w];
      before += w < wv ? cnt : 0;
      total += cnt;
    }
    if (valid) slist[before + __popcll(bal & ((1ull << lane) - 1ull))] = i32x4_t{d.qsb, d.meta, d.base, d.srow};
    const int n = __builtin_amdgcn_readfirstlane(total);
    if (tid < NST + 1) slist[n + tid] = i32x4_t{0, 0, 0, 0};     // what the ring reads past the end: nothing to load
    __syncthreads();
    return n;
  };
  auto entry = [&](int e) {
    const int4 a = scalarize(slist[e]);
    return QStage{a.x, a.y, a.z, a.w};
  };
  const int act_bit = 1 << (8 + wave), mask_bit = 1 << (12 + wave);

  // ---- LDS-DMA sources: descriptors over this batch row's slices, lane part of the offsets
  const size_t qrow_elems = (size_t)Nh * D;
  const uint32_t q_bytes = (uint32_t)min((size_t)qv.rpb * qrow_elems * 2, (size_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rq =
      __builtin_amdgcn_make_buffer_rsrc((void*)(Q + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdo =
      __builtin_amdgcn_make_buffer_rsrc((void*)(dO + (size_t)b * qv.rpb * qrow_elems), 0, q_bytes, 0x00020000);
  const uint32_t s_bytes = (uint32_t)((size_t)Nh * qv.rpb * 4);
  const __amdgpu_buffer_rsrc_t raux = __builtin_amdgcn_make_buffer_rsrc(
      wave == 0 ? (void*)(LSE2 + (size_t)b * Nh * qv.rpb)
                : wave == 1 ? (void*)(Delta + (size_t)b * Nh * qv.rpb) : (void*)(doc + (size_t)b * T),
      0, wave < 2 ? s_bytes : (uint32_t)T * 4, 0x00020000);
  // lane L of a piece writes LDS chunk L = (row L >> 2, physical chunk L & 3) of a 16-row x 64-byte panel slab
  const int rr = lane >> 2;
  const uint32_t voff = (uint32_t)(((size_t)rr * qrow_elems + 8 * ((lane & 3) ^ ((rr >> 2) & 3))) * 2);
  constexpr uint32_t OOB = 0x80000000u;       // >= num_records: the load returns 0 and touches no memory
  // Always IPS instructions per wave (the vmcnt arithmetic of the ring stays uniform): an invalid stage has left = 0,
  // every lane is out of range, nothing is read and the slot is filled with zeros nobody looks at.
  auto issue = [&](const QStage& d, int sbase) {
    char* st = smem + sbase;
    const int left = d.meta & 0xff;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave + 4 * i, panel = pc % Tile::NP, rh = pc / Tile::NP;
      const uint32_t vo = (16 * rh + rr < left) ? voff : OOB;
      const uint32_t so = (uint32_t)d.base + (uint32_t)((16 * rh * qrow_elems + 32 * panel) * 2);
      char* dst = st + panel * (Tile::PSTRIDE * 2) + rh * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)dst, 16, vo, so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdo, (lds_ptr_t)(dst + IMGB), 16, vo, so, 0, 0);
    }
    // aux row of this wave: 0 lse, 1 delta, 2 / 3 doc ids — ONE instruction, descriptor and offset picked by scalar selects
    const uint32_t va = lane < left ? (uint32_t)lane * 4 : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(raux, (lds_ptr_t)(st + 2 * IMGB + 256 * wave), 4, va,
                                             wave < 2 ? (uint32_t)d.srow : (uint32_t)d.qsb * 4, 0, 0);
  };

  static_for<2 * DBLK>([&](auto I) { acc_zero<decltype(I)::value>(); });

  const PRowReader<BQ, D> rrd(l31, hi);
  const PTrReader<BQ, D> trd(lane);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t trb0 = lds0 + 2 * (uint32_t)trd.t[0], trb1 = lds0 + 2 * (uint32_t)trd.t[1];
  const uint32_t rwb0 = lds0 + 2 * (uint32_t)rrd.a[0], rwb1 = lds0 + 2 * (uint32_t)rrd.a[1];
  const uint32_t auxb = lds0 + 16 * (uint32_t)hi;

  // ---- state that travels from one loop trip to the next: stage c = the stage whose S / dP / P the previous trip made
  bf16x8_t Pc[2];           // P(c) packed (B operand of dV^T += dO^T P, contraction slots = query rows 16 sp ..)
  float pf[16];             // P(c) in fp32
  f32x16_t dPc;             // dP(c)
  bool act_c = false;       // this wave takes part in stage c

  // One loop trip.  CUR: the wave has a stage c to finish (dV, dS, dK); NEXT: it starts stage n (S, dP, P); MASK: stage n
  // needs the element-wise predicate.  sb_c / sb_n: byte offsets of the two slots; the stage three ahead goes to sb_a.
  //
  // EVERY LDS read of the trip is inline asm and is retired by a hand-counted `s_waitcnt lgkmcnt(k)` that names the
  // registers it releases (LDS operations retire in order, so k = the number of LDS operations issued behind the one
  // needed).  Compiler-visible loads next to asm loads get waits computed from hipcc's own count, which ignores the asm
  // ones: correct but far too strong (lgkmcnt(3) in front of every S MFMA drained the transpose reads issued one gap
  // earlier).  LDS operations of the trip, in issue order (T = ds_read_b64_tr_b16, everything else ds_read_b128):
  //   head      16 T (dO^T operands of group 1), DE0, DE1                      retired by lgkmcnt(0) behind the barrier
  //   group 1   gap 2: DE2   gap 3: DE3   gaps 4..7: RQ0..RQ3
  //   group 2   gap s < 4: RQ[s + 4], T, T, T    gaps 4, 5: T, T, LE[s - 4] (, QD)    gaps 6, 7: LE[s - 4] (, QD)
  //             in front of S MFMA s: lgkmcnt(g2_wait(s))                      lgkmcnt(0) in front of group 3
  //   group 3   gap i: RDO[i]
  //   group 4   in front of dP MFMA s: lgkmcnt(7 - s)
  // The counts hold for CUR && NEXT; any other variant waits with lgkmcnt(0) (those trips are the rare ones).
  auto trip = [&](auto cur_t, auto next_t, auto mask_t, int qsb_n, const QStage& ahead, int sb_c, int sb_n,
                  int sb_a) {
    constexpr bool CUR = decltype(cur_t)::value, NEXT = decltype(next_t)::value, MASK = decltype(mask_t)::value;
    constexpr bool BOTH = CUR && NEXT;
    u32x2_t th[8][2];
    bf16x8_t tf[8];
    // (element reads of these go through float / int vectors: `__builtin_bit_cast(float, v[i])` on an element of an
    // integer vector reads element 0 whatever i is — a vector element has no address of its own)
    f32x4_t de4[4], le4[4];
    i32x4_t qd4[4];
    u32x4_t rq_[8], rdo_[8];
    f32x16_t S, dPn;
    float pn[16], dsf[16];
    bf16x8_t Pn[2], dSc[2];
    const uint32_t ta0 = trb0 + (uint32_t)sb_c, ta1 = trb1 + (uint32_t)sb_c;      // transposed operands of slot c
    const uint32_t rn0 = rwb0 + (uint32_t)sb_n, rn1 = rwb1 + (uint32_t)sb_n;      // row operands of slot n
    const uint32_t axc = auxb + (uint32_t)sb_c, axn = auxb + (uint32_t)sb_n;      // aux rows (+ 16 hi) of both slots
    auto tr_retire = [&]() {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u32x4_t t = {th[i][0].x, th[i][0].y, th[i][1].x, th[i][1].y};
        tf[i] = __builtin_bit_cast(bf16x8_t, t);
      }
    };
    // ---- before the barrier: slot c is stable, its dO^T operands and delta rows can travel while the wave waits
    if constexpr (CUR) {
      static_for<8>([&](auto I) {
        constexpr int i = decltype(I)::value, sp = i >> 2, db = i & 3;
        constexpr int off = 2 * (Tile::SIZE + db * Tile::PSTRIDE + 16 * sp * 32);
        th[i][0] = ds_tr16<off>(ta0);
        th[i][1] = ds_tr16<off + 512>(ta1);
      });
      de4[0] = ds_b128<f32x4_t, 2 * IMGB + 256>(axc);
      de4[1] = ds_b128<f32x4_t, 2 * IMGB + 256 + 32>(axc);
    }
    // my pieces of stage n have landed (stage n + 1 may stay in flight) ...
    wait_vmcnt<IPS>();
    // ... everybody's have, and everybody has left stage c - 1: its slot takes the stage three ahead
    if constexpr ((TN_FKV_ABL & 2) == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr ((TN_FKV_ABL & 1) == 0) issue(ahead, sb_a);
    auto ds_elem = [&](auto R) {               // dS = P o (dP - delta), element R of stage c
      constexpr int r = decltype(R)::value;
      float v = (TN_FKV_ABL & 4) ? dPc[r] : pf[r] * (dPc[r] - de4[r >> 2][r & 3]);
      asm volatile("" : "+v"(v));              // (one element per statement: no v_pk_*_f32 packing beside the MFMAs)
      dsf[r] = v;
    };
    auto p_elem = [&](auto R) {                // P = exp2(S c - LSE2), element R of stage n
      constexpr int r = decltype(R)::value;
      float pv = (TN_FKV_ABL & 4) ? S[r] : fast_exp2(S[r] * scale_log2 - le4[r >> 2][r & 3]);
      if constexpr (MASK) {
        const int o = 8 * (r >> 2) + 4 * hi + (r & 3);
        pv = ((kvrow <= qsb_n + o) & (qd4[r >> 2][r & 3] == dkdoc) & (dkdoc > 0)) ? pv : 0.f;
      }
      // (the empty statement keeps the element HERE: without it the optimiser sinks the whole softmax behind the loop
      // trip's last MFMA, next to the only use of P(n) — the next trip)
      asm volatile("" : "+v"(pv));
      pn[r] = pv;
    };
    auto pack8 = [&](const float (&p)[16], int sp) {
      const u32x4_t t = {pack2bf(p[8 * sp + 0], p[8 * sp + 1]), pack2bf(p[8 * sp + 2], p[8 * sp + 3]),
                         pack2bf(p[8 * sp + 4], p[8 * sp + 5]), pack2bf(p[8 * sp + 6], p[8 * sp + 7])};
      return __builtin_bit_cast(bf16x8_t, t);
    };
    // ---- group 1: dV^T(c) += dO^T(c) P(c)      gaps: first half of dS(c); delta rows 2, 3; Q(n) row operands 0..3
    if constexpr (CUR) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(th[0][0]), "+v"(th[0][1]), "+v"(th[1][0]), "+v"(th[1][1]), "+v"(th[2][0]), "+v"(th[2][1]),
                     "+v"(th[3][0]), "+v"(th[3][1]), "+v"(th[4][0]), "+v"(th[4][1]), "+v"(th[5][0]), "+v"(th[5][1]),
                     "+v"(th[6][0]), "+v"(th[6][1]), "+v"(th[7][0]), "+v"(th[7][1]), "+v"(de4[0]), "+v"(de4[1]));
      tr_retire();
    }
    static_for<8>([&](auto I) {
      constexpr int i = decltype(I)::value, sp = i >> 2, db = i & 3;
      if constexpr (CUR) {
        acc_mfma<db, i == 0>(tf[i], Pc[sp]);
        ds_elem(std::integral_constant<int, i>{});
        if constexpr (i == 2 || i == 3) de4[i] = ds_b128<f32x4_t, 2 * IMGB + 256 + 32 * i>(axc);
      }
      if constexpr (NEXT && i >= 4) {
        constexpr int s = i - 4;
        rq_[s] = ds_b128<u32x4_t, 2 * ((s >> 1) * Tile::PSTRIDE)>((s & 1) ? rn1 : rn0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- group 2: S(n) = Q(n) K^T               gaps: second half of dS(c); Q row operands 4..7; the Q^T(c) operands
    //                                                   of group 3; LSE / document-id rows of stage n
    if constexpr (CUR) dSc[0] = pack8(dsf, 0);
    if constexpr (CUR && !NEXT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(de4[2]), "+v"(de4[3]));
    static_for<8>([&](auto I) {
      constexpr int s = decltype(I)::value;
      if constexpr (NEXT) {
        constexpr int k = BOTH ? g2_wait(s, MASK) : 0;
        if constexpr (s == 0 && CUR)            // (the delta rows 2, 3 are older than Q row operand 0)
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(de4[2]), "+v"(de4[3]) : "n"(k));
        if constexpr (s == 0) mfma_first<k>(S, rq_[0], kreg[0]);
        else mfma_acc<k>(S, rq_[s], kreg[s]);
        if constexpr (s < 4) rq_[s + 4] = ds_b128<u32x4_t, 2 * (((s + 4) >> 1) * Tile::PSTRIDE)>((s & 1) ? rn1 : rn0);
      }
      if constexpr (CUR) {
        ds_elem(std::integral_constant<int, 8 + s>{});
        if constexpr (s < 6) {                  // 16 transpose reads in the first six gaps (3, 3, 3, 3, 2, 2)
          constexpr int first = s < 4 ? 3 * s : 12 + 2 * (s - 4), cnt = s < 4 ? 3 : 2;
          static_for<cnt>([&](auto J) {
            constexpr int x = first + decltype(J)::value, i = x >> 1, half = x & 1;
            constexpr int off = 2 * ((i & 3) * Tile::PSTRIDE + 16 * (i >> 2) * 32) + 512 * half;
            th[i][half] = ds_tr16<off>(half ? ta1 : ta0);
          });
        }
      }
      if constexpr (NEXT && s >= 4) {
        le4[s - 4] = ds_b128<f32x4_t, 2 * IMGB + 32 * (s - 4)>(axn);
        if constexpr (MASK) qd4[s - 4] = ds_b128<i32x4_t, 2 * IMGB + 512 + 32 * (s - 4)>(axn);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- group 3: dK^T(c) += Q^T(c) dS(c)       gaps: first half of P(n); the dO(n) row operands
    if constexpr (CUR) dSc[1] = pack8(dsf, 1);
    if constexpr (CUR && NEXT && MASK) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(th[0][0]), "+v"(th[0][1]), "+v"(th[1][0]), "+v"(th[1][1]), "+v"(th[2][0]), "+v"(th[2][1]),
                     "+v"(th[3][0]), "+v"(th[3][1]), "+v"(th[4][0]), "+v"(th[4][1]), "+v"(th[5][0]), "+v"(th[5][1]),
                     "+v"(th[6][0]), "+v"(th[6][1]), "+v"(th[7][0]), "+v"(th[7][1]), "+v"(le4[0]), "+v"(le4[1]),
                     "+v"(le4[2]), "+v"(le4[3]), "+v"(qd4[0]), "+v"(qd4[1]), "+v"(qd4[2]), "+v"(qd4[3]));
    } else if constexpr (CUR && NEXT) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(th[0][0]), "+v"(th[0][1]), "+v"(th[1][0]), "+v"(th[1][1]), "+v"(th[2][0]), "+v"(th[2][1]),
                     "+v"(th[3][0]), "+v"(th[3][1]), "+v"(th[4][0]), "+v"(th[4][1]), "+v"(th[5][0]), "+v"(th[5][1]),
                     "+v"(th[6][0]), "+v"(th[6][1]), "+v"(th[7][0]), "+v"(th[7][1]), "+v"(le4[0]), "+v"(le4[1]),
                     "+v"(le4[2]), "+v"(le4[3]));
    } else if constexpr (CUR) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(th[0][0]), "+v"(th[0][1]), "+v"(th[1][0]), "+v"(th[1][1]), "+v"(th[2][0]), "+v"(th[2][1]),
                     "+v"(th[3][0]), "+v"(th[3][1]), "+v"(th[4][0]), "+v"(th[4][1]), "+v"(th[5][0]), "+v"(th[5][1]),
                     "+v"(th[6][0]), "+v"(th[6][1]), "+v"(th[7][0]), "+v"(th[7][1]));
    } else if constexpr (NEXT && MASK) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(le4[0]), "+v"(le4[1]), "+v"(le4[2]), "+v"(le4[3]), "+v"(qd4[0]),
                   "+v"(qd4[1]), "+v"(qd4[2]), "+v"(qd4[3]));
    } else if constexpr (NEXT) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(le4[0]), "+v"(le4[1]), "+v"(le4[2]), "+v"(le4[3]));
    }
    if constexpr (CUR) tr_retire();
    if constexpr (NEXT && !CUR) mfma_result_pad(S);      // (with CUR the first MFMAs of this group are the distance)
    static_for<8>([&](auto I) {
      constexpr int i = decltype(I)::value, sp = i >> 2, db = i & 3;
      if constexpr (CUR) acc_mfma<DBLK + db, i == 0>(tf[i], dSc[sp]);
      if constexpr (NEXT) {
        if constexpr (i == 0 && CUR) asm volatile("s_nop 4" : "+v"(S));   // S chain -> first VALU read: >= 13 states
        p_elem(std::integral_constant<int, i>{});
        rdo_[i] = ds_b128<u32x4_t, 2 * (Tile::SIZE + (i >> 1) * Tile::PSTRIDE)>((i & 1) ? rn1 : rn0);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // ---- group 4: dP(n) = dO(n) V^T             gaps: second half of P(n)
    if constexpr (NEXT) {
      Pn[0] = pack8(pn, 0);
      asm volatile("" : "+v"(Pn[0]));
    }
    static_for<8>([&](auto I) {
      constexpr int s = decltype(I)::value;
      if constexpr (NEXT) {
        if constexpr (s == 0) mfma_first<7>(dPn, rdo_[0], vreg[0]);
        else mfma_acc<7 - s>(dPn, rdo_[s], vreg[s]);
        p_elem(std::integral_constant<int, 8 + s>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (NEXT) {
      Pn[1] = pack8(pn, 1);
      asm volatile("" : "+v"(Pn[1]));
      mfma_result_pad(dPn);
      Pc[0] = Pn[0];
      Pc[1] = Pn[1];
      dPc = dPn;
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[r] = pn[r];
    }
  };

  for (int cb = 0; cb < total_c; cb += LCAP) {
    const int n = build_list(cb);
    // stage s of the chunk lives in slot (s + 1) & 3; trip `it` finishes stage it and starts stage it + 1
    {
      const QStage s0 = entry(0), s1 = entry(1);
      issue(s0, 1 * STAGEB);
      issue(s1, 2 * STAGEB);
    }
    // the trips only need {qsb, meta} of the stage they start; the full entry is needed once, for the DMA three stages ahead
    int q1 = __builtin_amdgcn_readfirstlane(slist[0].x), m1 = __builtin_amdgcn_readfirstlane(slist[0].y);
    int q2 = __builtin_amdgcn_readfirstlane(slist[1].x), m2 = __builtin_amdgcn_readfirstlane(slist[1].y);
    act_c = false;
    // the list entry of the stage three ahead is fetched one trip early (a dependent LDS read + readfirstlanes at the
    // head of every trip would sit on the one wave's critical path)
    i32x4_t e0 = slist[2];
    int it = -1, sb_c = 0, sb_n = 0, sb_a = 0;
    bool act_n = false, mask_n = false;
    QStage ahead;
    auto head = [&]() {                  // what trip `it` needs (SGPR values: scalar compares / branches)
      const int4 ea = scalarize(e0);
      ahead = QStage{ea.x, ea.y, ea.z, ea.w};
      e0 = slist[it + 4];
      const int sc = (it + 1) & 3;
      sb_c = sc * STAGEB;
      sb_n = ((sc + 1) & 3) * STAGEB;
      sb_a = ((sc + 3) & 3) * STAGEB;
      act_n = (m1 & act_bit) != 0;
      mask_n = (m1 & mask_bit) != 0;
    };
    auto tail = [&]() {
      act_c = act_n;
      q1 = q2;
      m1 = m2;
      q2 = ahead.qsb;
      m2 = ahead.meta;
      ++it;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    head();
    while (true) {
      if (act_c && act_n && !mask_n) {
        // The interior trips of a wave form their OWN loop: the state that travels between trips (P, dP of stage c) then
        // stays in one set of registers; as one of six variants of a common loop body hipcc copied it (~85 v_mov per
        // trip) at every join.  Every variant holds exactly one barrier, so waves may sit in different loops.
        do {
          trip(T_{}, T_{}, F_{}, q1, ahead, sb_c, sb_n, sb_a);
          tail();
          if (it >= n) break;
          head();
        } while (act_n && !mask_n);
        if (it >= n) break;
        continue;
      }
      if (act_c) {
        if (act_n) trip(T_{}, T_{}, T_{}, q1, ahead, sb_c, sb_n, sb_a);
        else trip(T_{}, F_{}, F_{}, q1, ahead, sb_c, sb_n, sb_a);
      } else {
        if (act_n) {
          if (mask_n) trip(F_{}, T_{}, T_{}, q1, ahead, sb_c, sb_n, sb_a);
          else trip(F_{}, T_{}, F_{}, q1, ahead, sb_c, sb_n, sb_a);
        } else {
          trip(F_{}, F_{}, F_{}, q1, ahead, sb_c, sb_n, sb_a);
        }
      }
      tail();
      if (it >= n) break;
      head();
    }
    wait_vmcnt<0>();      // (the zero-fill tail DMAs must not land on the next chunk's stages)
    __syncthreads();
  }

  asm volatile("s_nop 15\n\ts_nop 15");     // the last MFMAs have written their accumulators
  // ---- epilogue: the ring is quiet behind the last chunk's barrier; each wave writes its 32 x D blocks of dV and dK
  // (8-byte runs of the accumulator layout) into private images and reads them back as whole rows — 16-byte stores, four
  // rows per instruction, instead of 8 bytes per lane at a row stride (attn_fwd_stream.hip)
  {
    constexpr int CPR = D / 8, RPI = 64 / CPR, NI = 32 / RPI;      // 16-byte chunks per row, rows per store instruction
    const int cc = lane % CPR, r0 = lane / CPR;
    // rcos / rsin (tn_attn_bwd_rope): dK leaves as the gradient of the UN-rotated k; table entries asked for up front
    // (attn_bwd_dq_stream.hip's epilogue)
    const bool rot = rcos != nullptr;
    u32x4_t c4[NI] = {}, s4[NI] = {};
    if (rot) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const size_t to = ((size_t)b * T + min(wk0 + i * RPI + r0, T - 1)) * (D / 2) + (cc & (CPR / 2 - 1)) * 8;
        c4[i] = *reinterpret_cast<const u32x4_t*>(rcos + to);
        s4[i] = *reinterpret_cast<const u32x4_t*>(rsin + to);
      }
    }
    char* obv = smem + wave * (2 * 32 * OSTR);
    char* obk = obv + 32 * OSTR;
    static_for<DBLK>([&](auto DB) {
      constexpr int db = decltype(DB)::value;
      static_for<4>([&](auto R4) {
        constexpr int r4 = decltype(R4)::value;
        u32x2_t o;
        o.x = pack2bf(acc_read<16 * db + 4 * r4 + 0>(), acc_read<16 * db + 4 * r4 + 1>());
        o.y = pack2bf(acc_read<16 * db + 4 * r4 + 2>(), acc_read<16 * db + 4 * r4 + 3>());
        *reinterpret_cast<u32x2_t*>(obv + l31 * OSTR + (32 * db + 8 * r4 + 4 * hi) * 2) = o;
        o.x = pack2bf(acc_read<64 + 16 * db + 4 * r4 + 0>() * scale, acc_read<64 + 16 * db + 4 * r4 + 1>() * scale);
        o.y = pack2bf(acc_read<64 + 16 * db + 4 * r4 + 2>() * scale, acc_read<64 + 16 * db + 4 * r4 + 3>() * scale);
        *reinterpret_cast<u32x2_t*>(obk + l31 * OSTR + (32 * db + 8 * r4 + 4 * hi) * 2) = o;
      });
    });
    const size_t off = (((size_t)b * T + wk0) * Nkv + hk) * D + cc * 8;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = i * RPI + r0;
      const u32x4_t v4 = *reinterpret_cast<const u32x4_t*>(obv + row * OSTR + cc * 16);
      u32x4_t k4 = *reinterpret_cast<const u32x4_t*>(obk + row * OSTR + cc * 16);
      if (rot) {
        const u32x4_t p4 = *reinterpret_cast<const u32x4_t*>(obk + row * OSTR + (cc ^ (CPR / 2)) * 16);
        k4 = rope_grad_chunk(k4, p4, c4[i], s4[i], cc >= CPR / 2);
      }
      if (wk0 + row < T) {
        *reinterpret_cast<u32x4_t*>(dV + off + (size_t)row * Nkv * D) = v4;
        *reinterpret_cast<u32x4_t*>(dK + off + (size_t)row * Nkv * D) = k4;
      }
    }
  }
}

// Launched by attn_bwd.hip's dispatcher for D = 128 (grid: kv heads x 128-row kv tiles x batch, 256 threads).
void launch_attn_bwd_kv_fused128(const bf16_t* Q, const bf16_t* K, const bf16_t* V, const bf16_t* dO,
                                 const float* lse2, const float* delta, bf16_t* dK, bf16_t* dV, const int* doc,
                                 AttnMeta m, QView qv, int B, int T, int Nh, int Nkv, float scale, float sl2,
                                 const bf16_t* rcos, const bf16_t* rsin, hipStream_t st) {
  dim3 gk(Nkv, (T + 127) / 128, B), block(256);
  hipLaunchKernelGGL((attn_bwd_kv_fused_kernel<128>), gk, block, 0, st, Q, K, V, dO, lse2, delta, dK, dV, doc, m, qv,
                     T, Nh, Nkv, scale, sl2, rcos, rsin);
}

}  // namespace tn
