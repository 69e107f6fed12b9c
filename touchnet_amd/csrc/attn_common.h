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
#include <type_traits>

#include "common.h"

namespace tn {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// LDS reads in kernels that fill LDS by LDS-DMA (buffer_load ... lds) must go through these ext_vector_type pointers,
// NOT through HIP's uint4 / int4 / float4 structs: hipcc's waitcnt insertion puts an `s_waitcnt vmcnt(0)` in front of
// any LDS access that carries no alias metadata while an LDS-DMA is pending (it cannot tell the slots of a ring apart),
// which silently serialises a hand-counted vmcnt(N) pipeline; loads through vector-typed pointers carry TBAA metadata
// and are left alone (checked in the ISA: scripts/check_dma_waits.sh).

__device__ __forceinline__ f32x16_t mfma32(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// max of three: compiles to ONE v_max3_f32 when the file is built with -fno-honor-nans (touchnet_amd/build.py does
// that for the attention forward kernels; without the flag every MFMA output is canonicalised first: 2x the VALU
// work, still correct).  Deliberately NOT inline asm: hipcc's hazard recognizer does not look inside asm blocks, and
// a v_max3 that reads a just-issued MFMA's registers without the required wait states returns stale data —
// run-to-run 1-ulp differences in the D = 64 kernel were exactly that.
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
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
// Behind them (round 6), what a workgroup used to derive itself through dependent memory round trips at its start:
//   qstat [B, nq32, 4]   {min positive id, max id, 1 if a pad id occurs, 0} of every 32 positions (one wave's query rows)
//   klist [B, nq128, 4 + 4 * kListPre]   the KV tiles a 128-position CAUSAL query tile meets, all chunks:
//                        {count, query tile, 0, 0} then min(count, kListPre) entries {tile, min id, max id, min positive
//                        id}; count > kListPre: the kernel builds its list itself (long documents).
//                        (Records SORTED by count, heaviest workgroups first, were measured and dropped: the launch
//                        balances better — a scheduling model says 0.92 -> 0.97 of the CU slots busy — but neighbouring
//                        query tiles no longer run side by side, their shared K / V tiles fall out of the XCD's L2 and
//                        the wait for a tile grows by 40 %: 233 -> 242 us, profiles/r06a_*.)
//   qlist [B, nq128, 4 + 4 * kListPre]   the mirror image for the dK / dV pass: the 64-position QUERY tiles a 128-position
//                        KV tile meets under the causal mask, {count, kv tile, 0, 0} + entries {q tile, min id, max id,
//                        min positive id}
constexpr int kListPre = 64;
struct AttnMeta {
  const int* tmin;
  const int* tmax;
  const int* tminpos;
  const int* q_lo;
  const int* kv_hi;
  int nt;
  const int* qstat;
  const int* klist;
  int nq32, nq128;
  const int* qlist;
};
__host__ inline int attn_meta_ints(int B, int T) {
  const int nt = (T + kTile - 1) / kTile, nq32 = (T + 31) / 32, nq128 = (T + 127) / 128;
  return 5 * B * nt + 4 * B * nq32 + 2 * (4 + 4 * kListPre) * B * nq128;
}
__host__ inline AttnMeta make_attn_meta(const int* meta, int B, int T) {
  const int nt = (T + kTile - 1) / kTile, n = B * nt, nq32 = (T + 31) / 32, nq128 = (T + 127) / 128;
  return AttnMeta{meta, meta + n, meta + 2 * n, meta + 3 * n, meta + 4 * n, nt,
                  meta + 5 * n, meta + 5 * n + 4 * B * nq32, nq32, nq128,
                  meta + 5 * n + 4 * B * nq32 + (4 + 4 * kListPre) * B * nq128};
}

// Which rows of the (possibly sequence-sharded) query-side buffers a launch covers.  Q / O / dO / dQ are
// [B, rpb, Nh, D] and LSE / delta are [B, Nh, rpb]; segment s = local rows [row0, row0 + rows) holding GLOBAL
// positions [off, off + rows) of the packed row (K / V / doc ids / metadata are always global, [B, T, ...]).
// Plain attention: one segment {0, T, 0}, rpb = T.  Context parallel with head/tail load balancing
// (touchnet/utils/distributed.py:292-315 -> torch's round-robin CP sharding): two segments per rank.
// Segment offsets and all but the last segment's length must be multiples of 128.
struct QView {
  int nseg;
  int row0[2], rows[2], off[2];
  int rpb;
  // optional restriction of the KEY side to a set of sequence chunks (context parallel: the forward runs once on the
  // rank's own chunks while the remote ones travel, once on the remote ones, and the two are merged by their LSE):
  // kv_tpc = 64-position tiles per chunk (0 = no restriction), bit c of kv_mask = chunk c takes part
  int kv_tpc;
  unsigned long long kv_mask;
  // 0 = causal inside a document (every decoder; Qwen2-Audio's tower), 1 = BIDIRECTIONAL inside a document (the Whisper
  // speech encoder of Kimi-Audio: transformers' WhisperEncoder has no causal mask,
  // touchnet/models/kimi_audio/modeling_kimi_audio.py:933-960): the key range of a query tile then runs to the LAST
  // tile that shares a document with it and the predicate loses its `kv <= q` term
  int bidir;
  __device__ __forceinline__ bool kv_tile_on(int j) const { return kv_tpc == 0 || ((kv_mask >> (j / kv_tpc)) & 1ull); }
  __host__ __device__ int tiles(int s, int bm) const { return s < nseg ? (rows[s] + bm - 1) / bm : 0; }
  // tile `idx` of size bm over all segments -> local first row, global first position, rows left in segment
  __device__ __forceinline__ void tile(int idx, int bm, int& l0, int& g0, int& left) const {
    const int n0 = tiles(0, bm);
    const int s = idx >= n0 ? 1 : 0;
    const int lt = idx - (s ? n0 : 0);
    l0 = row0[s] + lt * bm;
    g0 = off[s] + lt * bm;
    left = rows[s] - lt * bm;
  }
};

// Workgroup -> (head, tile) mapping shared by the attention kernels.  Launch grids are (heads, tiles, batch):
//  * blockIdx.x = head slot.  Workgroups go to XCDs round-robin by linear id (MI355X_MICROARCH.md "Workgroup
//    dispatch"), so with the TILE index in x (the first version) XCD k received tiles {k, k+8, ..}: under a causal
//    mask XCD 7 had 21 % (256-row tiles) more work than the average and every XCD's L2 fetched the K/V of every
//    head.  With the head in x all tiles of a head run on ONE XCD: balanced work, K/V of a head read into one L2.
//    GQA: slot x -> head (x % Nkv) * G + x / Nkv, so that the G query heads of a kv head share its XCD(s).
//  * blockIdx.y = tile rank, heaviest first (q tiles: last rows first; kv tiles of dK/dV: first rows first), so
//    that the tail of the launch is made of the light workgroups.
__device__ __forceinline__ int head_of_slot(int x, int Nh, int Nkv) {
  const int G = Nh / Nkv;
  return (x % Nkv) * G + x / Nkv;
}

// Can any (q, kv) pair with q in a set having positive-id range [qminpos, qmax] and kv in tile j be
// allowed?  Conservative on purpose: false only when the id ranges are disjoint.
__device__ __forceinline__ bool tile_may_interact(int qminpos, int qmax, int kminpos, int kmax) {
  return !(qminpos == 0x7fffffff || kminpos == 0x7fffffff || kmax < qminpos || kminpos > qmax);
}

// ---- KV-tile list in LDS --------------------------------------------------------------------------------------
// The tile loops used to walk the metadata arrays themselves: per tile two or three DEPENDENT scalar global loads
// (interaction test of the next tile, then of the current one), ~1-2 k cycles of latency each way (s_memtime trace of
// the ping-pong forward: QK^T segment 2100 -> 860 cycles without them).  Instead the workgroup compacts, ONCE per
// chunk of tiles (the caller's list capacity), the tiles of [lo, hi] that may interact with its query id range into LDS:
//   entry = {tile, min id, max id, min positive id};  entries [n, n + 4) = sentinels {hi_all + 1, 0, 0, 0}.
// NT = threads per workgroup (all must call; contains barriers).  Returns n (wave-uniform, in an SGPR).
constexpr int kListCap = 1024;

template <int NT>
__device__ __forceinline__ int build_kv_list(int4* list, int* wcount, int lo, int hi, int sentinel, int bminpos,
                                             int bmax, const int* m_min, const int* m_max, const int* m_minpos,
                                             int tid, int kv_tpc = 0, unsigned long long kv_mask = ~0ull) {
  constexpr int NW = NT / 64;
  const int lane = tid & 63, wave = tid >> 6;
  int n = 0;
  for (int base = lo; base <= hi; base += NT) {
    const int j = base + tid;
    int mn = 0, mx = 0, mp = 0;
    bool ok = false;
    if (j <= hi) {
      mn = m_min[j];
      mx = m_max[j];
      mp = m_minpos[j];
      ok = tile_may_interact(bminpos, bmax, mp, mx) && (kv_tpc == 0 || ((kv_mask >> (j / kv_tpc)) & 1ull));
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int before = n, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int c = wcount[w];
      before += w < wave ? c : 0;
      total += c;
    }
    if (ok) list[before + __popcll(bal & ((1ull << lane) - 1ull))] = make_int4(j, mn, mx, mp);
    n += __builtin_amdgcn_readfirstlane(total);
    __syncthreads();
  }
  if (tid < 4) list[n + tid] = make_int4(sentinel, 0, 0, 0);
  __syncthreads();
  return n;
}

// list entry i as four scalars (one broadcast LDS read + 4 readfirstlane)
__device__ __forceinline__ int4 list_entry(const int4* list, int i) {
  const int4 e = list[i];
  return make_int4(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y),
                   __builtin_amdgcn_readfirstlane(e.z), __builtin_amdgcn_readfirstlane(e.w));
}
__device__ __forceinline__ int4 scalarize(i32x4_t e) {
  return make_int4(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y),
                   __builtin_amdgcn_readfirstlane(e.z), __builtin_amdgcn_readfirstlane(e.w));
}
__device__ __forceinline__ int4 scalarize(int4 e) {
  return make_int4(__builtin_amdgcn_readfirstlane(e.x), __builtin_amdgcn_readfirstlane(e.y),
                   __builtin_amdgcn_readfirstlane(e.z), __builtin_amdgcn_readfirstlane(e.w));
}

// max over the two 32-lane halves of a wave without an LDS round trip (ds_bpermute queues behind other waves'
// operand reads): v_permlane32_swap exchanges a's upper half with b's lower half.
__device__ __forceinline__ float half_swap_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// Positive-id range of a wave's 32 rows (both 32-lane halves hold the same rows), returned in SGPRs so that
// every tile decision derived from it is a scalar branch and not an exec-mask region.
__device__ __forceinline__ void wave_id_range(int id, int& minpos, int& mx) {
  int a = id > 0 ? id : 0x7fffffff, b = id;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    a = min(a, __shfl_xor(a, o, 64));
    b = max(b, __shfl_xor(b, o, 64));
  }
  minpos = __builtin_amdgcn_readfirstlane(a);
  mx = __builtin_amdgcn_readfirstlane(b);
}

// A wave-uniform predicate as a real scalar (SGPR) bool: `if (uniform(c))` compiles to s_cbranch_scc and both
// arms stay free of exec-mask bookkeeping.  (hipcc otherwise turns a mask-typed bool that it cannot prove
// uniform into one s_and_saveexec region PER ELEMENT of the unrolled softmax loops.)
__device__ __forceinline__ bool uniform(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }

// ---- Panel image: ONE LDS copy of an [R rows][D] bf16 tile that serves both MFMA operand shapes ------------------
// The tile is cut into D/32 column panels of [R][32] (64-byte rows, panel stride R*64 + 64 bytes); inside a row the
// four 16-byte chunks are permuted by XOR with (row >> 2) & 3:
//   element (r, d) lives at  (d >> 5) * PSTRIDE + r * 32 + 8 * (((d >> 3) & 3) ^ ((r >> 2) & 3)) + (d & 7)
// * "row" operand (lane = tile row, 8 consecutive d per contraction slot group): one ds_read_b128; the 16 lanes of a
//   b128 service group hold 16 distinct (r & 15) -> 16 distinct 16-byte slots of the 256-byte bank row;
// * "transposed" operand (lane = d, contraction slots = tile rows): two ds_read_b64_tr_b16 (gfx950 transpose read: the
//   16 lanes of a group each point at 4 consecutive d of one row, rows 4 apart in groups of 4; lane i of the group
//   receives column i of the [4 rows][16 d] block — measured with scripts/microbench/tr_read_probe.hip).  The 32 lanes
//   of a service group cover 4 consecutive rows x 64 bytes = one contiguous 256-byte bank row: conflict free;
// * fill: 16-byte stores; 8 consecutive lanes write chunks 8p..8p+7 of one row = two neighbouring panels, whose
//   addresses differ by 64 bytes mod 128: conflict free for ds_write_b128's 8-lane groups.
// Replaces the row-major + transposed image pairs (the transposed ones were written with 8-byte transposing stores:
// 16 v_perm per 4x8 unit and 24-36 % LDS bank-conflict cycles, profiles/r02_attention_pmc_final.md).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

template <int R, int D>
struct PTile {
  static constexpr int NP = D / 32, PSTRIDE = R * 32 + 32, SIZE = NP * PSTRIDE;   // elements
  static __device__ __forceinline__ int chunk_off(int r, int c) {                  // c = d >> 3
    return (c >> 2) * PSTRIDE + r * 32 + 8 * ((c & 3) ^ ((r >> 2) & 3));
  }
};

// "row" operand reads: zero address VALU inside the tile loops (lane part: two VGPRs; the rest is a compile-time offset)
template <int R, int D>
struct PRowReader {
  int a[2];
  __device__ __forceinline__ PRowReader(int l31, int hi) {
    const int x = (l31 >> 2) & 3;
    a[0] = l31 * 32 + 8 * (hi ^ x);
    a[1] = l31 * 32 + 8 * ((2 + hi) ^ x);
  }
  // contraction slots d = 16 * s + 8 * hi + (0..7) of row rb + l31 (rb a multiple of 16)
  __device__ __forceinline__ bf16x8_t operand(const bf16_t* img, int rb, int s) const {
    return __builtin_bit_cast(
        bf16x8_t, *reinterpret_cast<const u32x4_t*>(img + (s >> 1) * PTile<R, D>::PSTRIDE + rb * 32 + a[s & 1]));
  }
};

// "transposed" operand reads: lane = d (32 * db + (lane & 31)), slots (hi, i) = rows kb + 8 * (i >> 2) + 4 * hi + (i & 3)
// (kb a multiple of 16) — exactly the rows a lane of a 32x32 MFMA result holds in registers 8 * sp .. 8 * sp + 7.
template <int R, int D>
struct PTrReader {
  int t[2];
  __device__ __forceinline__ PTrReader(int lane) {
    const int s4 = lane & 15, half = (lane >> 4) & 1, hi = lane >> 5;
    const int j = s4 >> 2, q = 2 * half + ((s4 & 3) >> 1);
    t[0] = (4 * hi + j) * 32 + 8 * (q ^ hi) + 4 * (s4 & 1);
    t[1] = (4 * hi + j) * 32 + 8 * ((q ^ hi) ^ 2) + 4 * (s4 & 1);
  }
  static __device__ __forceinline__ s16x4_t tr(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
  }
  __device__ __forceinline__ bf16x8_t operand(const bf16_t* img, int db, int kb) const {
    const bf16_t* base = img + db * PTile<R, D>::PSTRIDE + kb * 32;
    const s16x4_t lo = tr(base + t[(kb >> 3) & 1]);
    const s16x4_t up = tr(base + 8 * 32 + t[((kb >> 3) & 1) ^ 1]);
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
  }
};

// (With LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of the transpose-read builtin as well — the builtin
// carries no alias metadata.  At the prefetch distance the dK/dV kernel can afford, one stage, that wait costs nothing;
// an inline-asm read with hand-placed lgkmcnt waits removes it and measured 1-5 % slower, so the builtin stays.)

// Wave-uniform buffer descriptor over the first `rows_valid` rows of a [rows][ld] bf16 tile starting at
// `base`: loads past the last valid row return 0 in hardware (no exec-mask branches around tile edges).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const bf16_t* base, size_t ld, int rows_valid, int D) {
  const uint64_t p = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
  const uint32_t bytes = rows_valid > 0 ? (uint32_t)(((size_t)(rows_valid - 1) * ld + D) * 2) : 0u;
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Stage a [R rows][D] bf16 tile (source row stride `ld` elements, rows >= rows_valid zero-filled) into a PTile image:
// full-row coalesced 16-byte global loads now, 16-byte LDS stores later.
template <int R, int D, int NT>
struct PStage {
  static constexpr int CPR = D / 8;
  static constexpr int N = (R * CPR + NT - 1) / NT;
  static constexpr bool EXACT = (R * CPR) % NT == 0;
  uint4 v[N];
  __device__ __forceinline__ void load(const bf16_t* src, size_t ld, int rows_valid, int tid) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(src, ld, rows_valid < R ? rows_valid : R, D);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + i * NT;
      const int r = c / CPR, cc = c % CPR;
      v[i] = buf_load16(rs, (EXACT || c < R * CPR) ? (uint32_t)((r * ld + cc * 8) * 2) : 0xffffffffu);
    }
  }
  __device__ __forceinline__ void store(bf16_t* dst, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + i * NT;
      const int r = c / CPR, cc = c % CPR;
      if (EXACT || c < R * CPR) *reinterpret_cast<uint4*>(dst + PTile<R, D>::chunk_off(r, cc)) = v[i];
    }
  }
};

// The TRANSPOSED rotary rotation — what tn_rope_apply(backward = 1) applies to the gradient of a rotated q / k row — of one
// 16-byte chunk of a gradient row in a backward kernel's epilogue: `own` = the lane's 8 columns, `other` = the same 8
// columns of the row's other half, c4 / s4 = the 8 table entries of the row (bf16 [rows, D / 2] tables of tn_rope_table),
// upper = the lane's columns are >= D / 2.  rope_rotate itself on the bf16-rounded values: the bits the row kernel produces.
__device__ __forceinline__ u32x4_t rope_grad_chunk(u32x4_t own, u32x4_t other, u32x4_t c4, u32x4_t s4, bool upper) {
  // rope_rotate(a, b, c, -sin): lower half ya = fma(-b, -sin, a c) = fma(other, sin, own c); upper half yb = fma(a, -sin, b c)
  // = fma(other, -sin, own c) — sign flips of a factor are exact, so ONE form with the table's sign bits flipped for the upper
  // lanes gives the row kernel's bits at a third of its instructions (an epilogue's VALU time is exposed: 1-2 waves per SIMD)
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  const uint32_t flip = upper ? 0x80008000u : 0u;
  u32x4_t out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t sj = s4[j] ^ flip;
    const f32x2 x = {__uint_as_float(own[j] << 16), __uint_as_float(own[j] & 0xffff0000u)};
    const f32x2 y = {__uint_as_float(other[j] << 16), __uint_as_float(other[j] & 0xffff0000u)};
    const f32x2 c = {__uint_as_float(c4[j] << 16), __uint_as_float(c4[j] & 0xffff0000u)};
    const f32x2 sv = {__uint_as_float(sj << 16), __uint_as_float(sj & 0xffff0000u)};
    const f32x2 t = x * c;
    const f32x2 r = __builtin_elementwise_fma(y, sv, t);
    out[j] = pack2bf(r.x, r.y);
  }
  return out;
}

}  // namespace tn
