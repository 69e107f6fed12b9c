// Hand-written bf16 MFMA GEMM for gfx950 (MI355X): the linear layers of the decoder / audio-tower blocks.
//
//     C[M, N] (bf16) = A[M, K] · B[N, K]^T  (+ bias[N]) (+ C)        fp32 accumulation, one rounding
//
// Both operands are contraction-contiguous ("TN", nn.Linear's forward layout: x [tokens, in] · W [out, in]^T) —
// the layout every GEMM of the step is brought into.  Replaces the projections the reference
// leaves to torch / liger (q/k/v/o/gate/up/down/lm_head of transformers' LlamaMLP / LlamaAttention as swapped at
// touchnet/models/llama/__init__.py:11-15).
//
// Structure (MI355X-first; numbers from MI355X_MICROARCH.md):
//  * 256 x 256 output tile per 512-thread workgroup (8 waves = 2(M) x 4(N), wave tile 128 x 64 = 4 x 2 blocks of
//    v_mfma_f32_32x32x16_bf16, 128 accumulator registers), one workgroup per CU, two waves per SIMD.
//  * K is consumed in 64-deep stages ([256 rows][128 B] per operand = full cache lines) through LDS filled by
//    LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass).  The DMA image is lane-linear, so
//    the bank swizzle is applied to the per-lane SOURCE address and undone by the same XOR on the ds_read_b128 side:
//    16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7) of the row's 128 bytes — conflict-free for the b128
//    lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (measured: SQ_LDS_BANK_CONFLICT = 1 % of LDS cycles).
//  * ONE barrier per stage, placed in front of the stage's LAST 16-deep quarter: fragments of quarter q+1 are read
//    while quarter q's 8 MFMAs run, so the matrix pipe does not drain across the barrier, and never a vmcnt(0) in the
//    loop (see the two main loops below for the slot accounting).
//  * Out-of-range rows (M, N not multiples of 256) are zero-filled by the buffer descriptor's bounds check; the
//    tail stages re-read stage 0 into dead slots so that the vmcnt arithmetic stays uniform.
//  * Workgroup -> tile map is XCD-aware: workgroup ids go to the 8 XCDs round-robin, so XCD x is given a contiguous
//    range of tiles walked in 8(M)-tall column-major groups: the 32 workgroups an XCD runs at once cover an
//    8 x 4 patch of tiles and share their A / B panels in that XCD's L2.
//  * Epilogue through LDS (the ring is dead by then): each wave parks its 128 x 64 bf16 tile in its own 16 KB,
//    XOR-swizzled, and writes full 128-byte lines with 16-byte stores (and, optionally, the TRANSPOSED tile for the
//    weight-gradient GEMM that consumes this output next — the standalone transpose pass disappears).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace tn {
namespace gemm {

constexpr int BM = 256, BN = 256;
constexpr int LDS_BYTES = 160 * 1024;                // ring5 uses all of it; the other loops and the epilogue 128 KB
constexpr int NT = 512;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Params {
  const bf16_t* A;
  const bf16_t* B;
  bf16_t* C;
  bf16_t* Ct;          // optional transposed copy [N, M] (ldct), or null
  const bf16_t* bias;  // optional [N]
  int M, N, K;
  long long lda, ldb, ldc, ldct;
  int accumulate;      // C += result (the group-accumulating input-gradient GEMMs)
  int nbm, nbn;
};

__device__ __forceinline__ bf16x8_t lds_frag(const char* smem, int byte_off) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(smem + byte_off));
}

// XCD-aware, bijective workgroup -> tile map
__device__ __forceinline__ void tile_of_block(int bid, int nbm, int nbn, int& tm, int& tn) {
  const int total = nbm * nbn;
  const int xcd = bid & 7, local = bid >> 3;
  const int q = total >> 3, r = total & 7;
  const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
#ifndef TN_GEMM_GM
#define TN_GEMM_GM 8
#endif
  constexpr int GM = TN_GEMM_GM;
  const int per_group = GM * nbn;
  const int g = vid / per_group, w = vid - g * per_group;
  const int gm = min(GM, nbm - g * GM);  // rows in this (possibly short, last) group
  tm = g * GM + w % gm;
  tn = w / gm;
}

// Everything a main loop needs to know about its workgroup / wave.
struct Ctx {
  char* smem;
  __amdgpu_buffer_rsrc_t ra, rb;
  long long lda2, ldb2;     // row pitches in bytes
  int wave, wr, wc, lane, l31, hi;
  int K;
};

typedef f32x16_t Acc[4][2];

// Timing experiments (scripts/build_variant.sh <name> -DTN_GEMM_ABLATE=n; results are garbage for n != 0):
//   1 no DMA in the loop   2 no fragment reads in the loop   3 no MFMA   4 no epilogue stores   5 no barrier in the loop
//   6 every DMA row reads row 0 (all L2 hits)
#ifndef TN_GEMM_ABLATE
#define TN_GEMM_ABLATE 0
#endif

// result rows (registers) = n, result column (lane) = m: 4 consecutive n per register quad -> 8-byte packs
__device__ __forceinline__ void mma1(Acc& acc, const bf16x8_t (&a)[4], const bf16x8_t (&b)[2], int i, int j) {
#if TN_GEMM_ABLATE == 3
  asm volatile("" ::"v"(a[i]), "v"(b[j]));
#else
  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Main loop A (reference for A/B runs, TN_GEMM_LOOP=1): two slots of 64-deep stages ([256 rows][128 B] per operand per slot = full cache lines: a DMA piece is
// 8 rows x 128 B), ONE barrier per 64 of K.  A stage is consumed in four 16-deep quarters; fragments of quarter q+1
// are read while quarter q's 8 MFMAs run.  The barrier sits in front of the LAST quarter of stage t: by then every
// wave has read all of stage t (its slot is free for stage t+2, issued right behind the barrier, piece by piece
// between MFMAs) and has waited for its own pieces of stage t+1 (issued one stage earlier), so the first quarter of
// stage t+1 is read under the last quarter's MFMAs and the matrix pipe never drains.
// Swizzle: 16-byte chunk c (0..7) of row r sits in slot c ^ ((r >> 1) & 7); a 256-byte bank row holds 2 tile rows.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mainloop_pair64(const Ctx& c, Acc& acc) {
  char* const smem = c.smem;
  constexpr int SLOT = 32768;                 // per operand per slot
  constexpr int LA = 0, LB = 2 * SLOT;
  const int wave = c.wave, lane = c.lane, l31 = c.l31, hi = c.hi;
  int voff_a[4], voff_b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {               // this wave stages rows [32 w, 32 w + 32) of both operands: 4 pieces each
    const int row = wave * 32 + q * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
#if TN_GEMM_ABLATE == 6
    voff_a[q] = chunk * 16;
    voff_b[q] = chunk * 16;
#else
    voff_a[q] = (int)(row * c.lda2) + chunk * 16;
    voff_b[q] = (int)(row * c.ldb2) + chunk * 16;
#endif
  }
  const int np = c.K / 64;
  auto issue_piece = [&](int slot, int soff, int piece) {          // piece 0..3 = A, 4..7 = B
    char* d = smem + (piece < 4 ? LA : LB) + slot * SLOT + wave * 4096 + (piece & 3) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(piece < 4 ? c.ra : c.rb, (lds_ptr_t)d, 16,
                                             piece < 4 ? voff_a[piece & 3] : voff_b[piece & 3], soff, 0, 0);
  };
  auto stage_soff = [&](int stage) { return stage < np ? stage * 128 : 0; };
  const int f = (l31 >> 1) & 7;
  int xo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) xo[q] = ((2 * q + hi) ^ f) << 4;
  const int a_base = LA + (c.wr * 128 + l31) * 128;
  const int b_base = LB + (c.wc * 64 + l31) * 128;
  bf16x8_t ae[4], be[2], ao[4], bo[2];
  auto read_q = [&](int slot, int q, bf16x8_t (&a)[4], bf16x8_t (&b)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = lds_frag(smem, b_base + slot * SLOT + j * 4096 + xo[q]);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = lds_frag(smem, a_base + slot * SLOT + i * 4096 + xo[q]);
  };
  auto mma8 = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mma1(acc, a, b, i, j);
  };

  {
    const int s0 = stage_soff(0), s1 = stage_soff(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(0, s0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(1, s1, q);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_q(0, 0, ae, be);
  __builtin_amdgcn_s_waitcnt(0xc07f);

  // hipcc's scheduler otherwise sinks the fragment reads down to their first use (shortest live ranges): the
  // software pipeline below is pinned group by group with sched_barrier(0)
#define TN_PIN() __builtin_amdgcn_sched_barrier(0)
#if TN_GEMM_ABLATE == 1
#define TN_LOOP_PIECE(slot, soff, q) asm volatile("" ::"s"(soff))
#else
#define TN_LOOP_PIECE(slot, soff, q) issue_piece(slot, soff, q)
#endif
#if TN_GEMM_ABLATE == 2
#define TN_LOOP_READ(slot, q, a, b) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]))
#else
#define TN_LOOP_READ(slot, q, a, b) read_q(slot, q, a, b)
#endif
  auto pair = [&](auto SLOTC, int t) {
    constexpr int slot = decltype(SLOTC)::value;
    const int soff = stage_soff(t + 2);
    __builtin_amdgcn_s_setprio(1);
    TN_LOOP_READ(slot, 1, ao, bo);
    TN_PIN();
    mma8(ae, be);
    TN_PIN();
    TN_LOOP_READ(slot, 2, ae, be);
    TN_PIN();
    mma8(ao, bo);
    TN_PIN();
    TN_LOOP_READ(slot, 3, ao, bo);
    TN_PIN();
    mma8(ae, be);
    TN_PIN();
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // my reads of stage t are complete (quarter 3 is in registers)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // my pieces of stage t+1 (the only ones in flight) have landed
#if TN_GEMM_ABLATE != 5
    __builtin_amdgcn_s_barrier();                       // stage t+1 visible; slot of stage t free
#endif
    __builtin_amdgcn_s_setprio(1);
    TN_LOOP_READ(slot ^ 1, 0, ae, be);
    TN_PIN();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mma1(acc, ao, bo, i, j);
        TN_PIN();
        TN_LOOP_PIECE(slot, soff, (i * 2 + j) / 2 + 4 * ((i * 2 + j) & 1));   // A0 B0 A1 B1 A2 B2 A3 B3
        TN_PIN();
      }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int t = 0; t < np; t += 2) {
    pair(std::integral_constant<int, 0>{}, t);
    pair(std::integral_constant<int, 1>{}, t + 1);
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Main loop B (default): the whole 160 KB of LDS as a ring of FIVE 32 KB operand slots (operand-stage j = A(t) for j = 2t,
// B(t) for j = 2t+1, slot j % 5), 64-deep stages in full cache lines like loop A.  Loop A can only start the DMA of
// stage t+2 once stage t is consumed and needs it one stage later: its 64 KB round trip (~2.7 k cycles measured with
// the MFMAs removed) is exposed whenever it exceeds a stage's MFMA time.  With the fifth slot 96 KB are in flight:
// at the barrier of stage t the freed slots take B(t+2) (needed one stage later, issued at once) and A(t+3) (needed
// TWO stages later, its pieces spread over the next stage's MFMAs); A(t+2) is already under way.
//   wait at the barrier of stage t: in flight, oldest first: A(t+1), B(t+1), A(t+2) -> vmcnt(4) = A(t+2) may remain.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mainloop_ring5(const Ctx& c, Acc& acc) {
  char* const smem = c.smem;
  constexpr int SLOT = 32768;
  const int wave = c.wave, lane = c.lane, l31 = c.l31, hi = c.hi;
  int voff_a[4], voff_b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wave * 32 + q * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voff_a[q] = (int)(row * c.lda2) + chunk * 16;
    voff_b[q] = (int)(row * c.ldb2) + chunk * 16;
  }
  const int np = c.K / 64;
  auto stage_soff = [&](int stage) { return stage < np ? stage * 128 : 0; };
  auto piece_a = [&](int slot, int soff, int q) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_ptr_t)(smem + slot * SLOT + wave * 4096 + q * 1024), 16,
                                             voff_a[q], soff, 0, 0);
  };
  auto piece_b = [&](int slot, int soff, int q) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rb, (lds_ptr_t)(smem + slot * SLOT + wave * 4096 + q * 1024), 16,
                                             voff_b[q], soff, 0, 0);
  };
  const int f = (l31 >> 1) & 7;
  int xa[4], xb[4];                           // per-quarter lane offsets inside a slot
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    xa[q] = (c.wr * 128 + l31) * 128 + (((2 * q + hi) ^ f) << 4);
    xb[q] = (c.wc * 64 + l31) * 128 + (((2 * q + hi) ^ f) << 4);
  }
  bf16x8_t ae[4], be[2], ao[4], bo[2];
  auto read_q = [&](int sa, int sb, int q, bf16x8_t (&a)[4], bf16x8_t (&b)[2]) {
    const int ba = sa * SLOT + xa[q], bb = sb * SLOT + xb[q];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = lds_frag(smem, bb + j * 4096);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = lds_frag(smem, ba + i * 4096);
  };
  auto mma8 = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) mma1(acc, a, b, i, j);
  };
  auto next = [](int s, int d) { s += d; return s >= 5 ? s - 5 : s; };

  // prologue: A0 B0 A1 B1 A2 -> slots 0..4
  {
    const int s0 = stage_soff(0), s1 = stage_soff(1), s2 = stage_soff(2);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(0, s0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b(1, s0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(2, s1, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b(3, s1, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(4, s2, q);
  }
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int sa = 0, sb = 1;                         // slots of A(t), B(t)
  read_q(sa, sb, 0, ae, be);
  __builtin_amdgcn_s_waitcnt(0xc07f);

  for (int t = 0; t < np; ++t) {
    const int sa1 = next(sa, 2), sb1 = next(sb, 2);       // slots of stage t+1
    const int soff_b = stage_soff(t + 2), soff_a = stage_soff(t + 3);
    __builtin_amdgcn_s_setprio(1);
    read_q(sa, sb, 1, ao, bo);
    TN_PIN();
    // quarter 0 (A(t+2)'s second half of pieces were issued in the previous trip's tail; see below)
    mma8(ae, be);
    TN_PIN();
    read_q(sa, sb, 2, ae, be);
    TN_PIN();
    mma8(ao, bo);
    TN_PIN();
    read_q(sa, sb, 3, ao, bo);
    TN_PIN();
    mma8(ae, be);
    TN_PIN();
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // my reads of stage t are complete
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // A(t+1), B(t+1) landed; A(t+2) may still be in flight
    __builtin_amdgcn_s_barrier();                       // stage t+1 visible; slots sa, sb free
    __builtin_amdgcn_s_setprio(1);
    read_q(sa1, sb1, 0, ae, be);
    TN_PIN();
    // last quarter of stage t: B(t+2) -> slot sa at once (due in one stage), then A(t+3) -> slot sb (due in two)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mma1(acc, ao, bo, i, j);
        TN_PIN();
        if (i < 2) piece_b(sa, soff_b, i * 2 + j);
        else piece_a(sb, soff_a, (i - 2) * 2 + j);
        TN_PIN();
      }
    __builtin_amdgcn_s_setprio(0);
    sa = sa1;
    sb = sb1;
  }
}

// Epilogue through LDS: the wave parks its 128 x (32 NJ) tile, 64 columns at a time, in its own XOR-swizzled LDS
// region ([128 rows][64 cols] bf16, 128-byte rows, chunk ^= row & 7) and writes full 128-byte lines.
template <bool HAS_CT, int NJ>
__device__ __forceinline__ void epilogue(const Params& p, f32x16_t (&acc)[4][NJ], char* park_base, int wm0, int wn0_base,
                                         int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dead tail DMAs must not land on the parked tile
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int half = 0; half < NJ / 2; ++half) {
  char* park = park_base + half * 16384;
  const int wn0 = wn0_base + half * 64;
  float bias_v[2][4][4];
  if (p.bias != nullptr) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = min(wn0 + j * 32 + 8 * g + 4 * hi, p.N - 4);    // (columns >= N are never stored)
        const uint2 w = *reinterpret_cast<const uint2*>(p.bias + n);
        bias_v[j][g][0] = __uint_as_float(w.x << 16);
        bias_v[j][g][1] = __uint_as_float(w.x & 0xffff0000u);
        bias_v[j][g][2] = __uint_as_float(w.y << 16);
        bias_v[j][g][3] = __uint_as_float(w.y & 0xffff0000u);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 32 + l31;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][half * 2 + j][4 * g + e] + (p.bias != nullptr ? bias_v[j][g][e] : 0.f);
        const int chunk = (j * 4 + g) ^ (row & 7);
        *reinterpret_cast<uint2*>(park + row * 128 + chunk * 16 + hi * 8) =
            make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
  }
  // (only this wave touches its park region: a wave-level wait is enough)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const bool acc_c = p.accumulate != 0;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int row = it * 8 + (lane >> 3), c = lane & 7;
    uint4 v = *reinterpret_cast<const uint4*>(park + row * 128 + ((c ^ (row & 7)) << 4));
    const int m = wm0 + row, n = wn0 + c * 8;
    if (m < p.M && n < p.N && TN_GEMM_ABLATE != 4) {     // N is a multiple of 8 (checked by the host)
      bf16_t* dst = p.C + (long long)m * p.ldc + n;
      if (acc_c) {
        Vec16<bf16_t> o, nw;
        o.load(dst);
        nw.raw = v;
        float fo[8], fn[8];
        o.unpack(fo);
        nw.unpack(fn);
#pragma unroll
        for (int e = 0; e < 8; ++e) fn[e] += fo[e];
        nw.pack(fn);
        v = nw.raw;
      }
      *reinterpret_cast<uint4*>(dst) = v;
    }
  }
  if constexpr (HAS_CT) {
    // transposed copy: Ct[n, m]; a lane gathers 8 consecutive m of one n from the parked tile (2-byte LDS reads:
    // this path trades LDS instructions for the HBM round trip of a separate transpose pass)
#pragma unroll 2
    for (int it = 0; it < 16; ++it) {
      const int n_l = it * 4 + (lane >> 4), mg = lane & 15;       // 64 n x 16 groups of 8 m
      uint32_t w[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        uint32_t lo, hi16;
        {
          const int row = mg * 8 + 2 * e2;
          lo = *reinterpret_cast<const uint16_t*>(park + row * 128 + ((((n_l >> 3) ^ (row & 7))) << 4) + (n_l & 7) * 2);
        }
        {
          const int row = mg * 8 + 2 * e2 + 1;
          hi16 = *reinterpret_cast<const uint16_t*>(park + row * 128 + ((((n_l >> 3) ^ (row & 7))) << 4) + (n_l & 7) * 2);
        }
        w[e2] = lo | (hi16 << 16);
      }
      const int n = wn0 + n_l, m = wm0 + mg * 8;
      if (n < p.N && m < p.M)     // M is a multiple of 8 when a transposed copy is requested (host check)
        *reinterpret_cast<uint4*>(p.Ct + (long long)n * p.ldct + m) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  }
}

template <bool HAS_CT, int LOOP>
__global__ __launch_bounds__(NT, 2) void gemm_tn_kernel(const Params p) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;   // 2 x 4 waves
  const int l31 = lane & 31, hi = lane >> 5;

  int tm, tn;
  tile_of_block(blockIdx.x, p.nbm, p.nbn, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA source descriptors: one per operand, based at the tile's first row (offsets stay < 2^31) ----------
  const long long a_left = (long long)(p.M - m0) * p.lda * 2, b_left = (long long)(p.N - n0) * p.ldb * 2;
  Ctx c;
  c.smem = smem;
  c.ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long long)m0 * p.lda), 0,
                                           (int)min(a_left, 0x7fffffffLL), 0x00020000);
  c.rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (long long)n0 * p.ldb), 0,
                                           (int)min(b_left, 0x7fffffffLL), 0x00020000);
  c.lda2 = p.lda * 2;
  c.ldb2 = p.ldb * 2;
  c.wave = wave; c.wr = wr; c.wc = wc; c.lane = lane; c.l31 = l31; c.hi = hi;
  c.K = p.K;

  Acc acc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef TN_GEMM_SKEW
  // experiment: de-phase the workgroups of an XCD (they all start together and would burst their DMA in lockstep)
  for (int i = ((blockIdx.x >> 3) & 31) * TN_GEMM_SKEW; i > 0; --i) __builtin_amdgcn_s_sleep(1);
#endif
  if constexpr (LOOP == 1)
    mainloop_pair64(c, acc);
  else
    mainloop_ring5(c, acc);

  epilogue<HAS_CT, 2>(p, acc, smem + wave * 16384, m0 + wr * 128, n0 + wc * 64, lane);
}

}  // namespace gemm
}  // namespace tn

extern "C" {

// C[M,N] = A[M,K] · B[N,K]^T (+ bias) (+ C if accumulate); optional transposed copy Ct[N,M].  bf16, fp32 accumulate.
// Requirements (else -22): K % 128 == 0, N % 8 == 0, lda/ldb/ldc % 8 == 0, 16-byte aligned bases;
// with Ct: M % 8 == 0, ldct % 8 == 0.
int tn_gemm_bf16_tn(const void* A, const void* B, void* C, void* Ct, const void* bias, int M, int N, int K,
                    long long lda, long long ldb, long long ldc, long long ldct, int accumulate, void* stream) {
  using namespace tn::gemm;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 128) != 0 || (N % 8) != 0) return TN_EINVAL;
  if ((lda % 8) || (ldb % 8) || (ldc % 8) || lda < K || ldb < K || ldc < N) return TN_EINVAL;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return TN_EINVAL;
  if (Ct != nullptr && ((M % 8) || (ldct % 8) || ldct < M || ((uintptr_t)Ct & 15))) return TN_EINVAL;
  if (Ct != nullptr && accumulate) return TN_EINVAL;
  // per-tile DMA offsets are 32-bit: 288 rows of the operand must stay below 2 GB
  if ((long long)288 * lda * 2 >= 0x7fffffffLL || (long long)288 * ldb * 2 >= 0x7fffffffLL) return TN_EINVAL;
  Params p;
  p.A = (const tn::bf16_t*)A;
  p.B = (const tn::bf16_t*)B;
  p.C = (tn::bf16_t*)C;
  p.Ct = (tn::bf16_t*)Ct;
  p.bias = (const tn::bf16_t*)bias;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldct = ldct;
  p.accumulate = accumulate;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (N + BN - 1) / BN;
  const dim3 grid(p.nbm * p.nbn), block(NT);
  hipStream_t st = (hipStream_t)stream;
  // TN_GEMM_LOOP=1: kernel-development A/B switch to the two-slot loop (default = the five-slot ring)
  static const int loop = [] { const char* e = getenv("TN_GEMM_LOOP"); return e ? atoi(e) : 2; }();
#define TN_LAUNCH(CT, L) hipLaunchKernelGGL((gemm_tn_kernel<CT, L>), grid, block, 0, st, p)
  if (Ct != nullptr) {
    if (loop == 1) TN_LAUNCH(true, 1); else TN_LAUNCH(true, 2);
  } else {
    if (loop == 1) TN_LAUNCH(false, 1); else TN_LAUNCH(false, 2);
  }
#undef TN_LAUNCH
  TN_LAUNCH_CHECK();
  return TN_OK;
}

}  // extern "C"
