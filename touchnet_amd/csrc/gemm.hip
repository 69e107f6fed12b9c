// Hand-written bf16 MFMA GEMM for gfx950 (MI355X): every product of the linear layers of the decoder / audio-tower
// blocks — forward, input gradient and weight gradient — in the layout the tensors already have.
//
//     C[M, N] (bf16) = sum over segments s of  opA_s · opB_s^T   (+ bias[N]) (+ C)       fp32 accumulation, one rounding
//
// Each operand of each segment is stored either
//   ROW    X[R, K]  contraction-contiguous (row pitch ld)   — x in y = x W^T, W in y = x W^T, dY in dX = dY W
//   KMAJ   X[K, R]  contraction-major      (row pitch ld)   — W in dX = dY W, dY and x in dW = dY^T x
// so the three GEMMs of a linear layer (transformers' LlamaMLP / LlamaAttention projections as swapped at
// touchnet/models/llama/__init__.py:11-15) read nn.Linear's own tensors: no transposed copies of
// W, dY or x are ever made (round 2 ran 480 standalone transpose launches per step for a ROW-only kernel).  Several
// segments = one accumulator over several (A, B) pairs: dX = dQ Wq + dK Wk + dV Wv is ONE launch without bf16 round trips.
//
// Structure (MI355X-first; numbers from MI355X_MICROARCH.md):
//  * 256 x 256 output tile per 512-thread workgroup (8 waves = 2(M) x 4(N), wave tile 128 x 64 = 4 x 2 blocks of
//    v_mfma_f32_32x32x16_bf16, 128 accumulator registers), one workgroup per CU, two waves per SIMD.
//  * K is consumed in 64-deep operand stages of 32 KB that reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
//    registers, no ds_write pass) in FULL 128-byte lines.  The DMA image is lane-linear, so every bank swizzle is
//    applied to the per-lane SOURCE address and undone on the read side:
//      ROW   image [256 r][128 B]: 16-byte chunk c of row r in slot c ^ ((r >> 1) & 7); fragment = one ds_read_b128
//      KMAJ  image [64 k][512 B] : 64-byte group g of row k in position g ^ (k & 3);    fragment = two
//            ds_read_b64_tr_b16 (gfx950 transpose read: the 16 lanes of a group point at 4 k-rows x 32 B and lane i
//            receives column i of that [4 k][16 r] block) — the four k-rows of a 32-lane service group sit in four
//            different 64-byte bank groups: conflict-free, same as the ROW image.
//    Both give a lane the contraction slots k = 8 * (lane >> 5) + (0..7) of row lane & 31, so the modes mix freely.
//  * 160 KB of LDS = ring of FIVE operand slots (operand-stage j = A(t) for j = 2t, B(t) for j = 2t+1, slot j % 5).
//    ONE barrier per stage, in front of its LAST 16-deep quarter (fragments of quarter q+1 are read while quarter q's
//    8 MFMAs run, so the matrix pipe does not drain across it).  The barrier of stage t frees two slots, which take
//    B(t+2) (due one stage later) and A(t+3) (due two stages later); their 8 pieces per wave are placed between the
//    MFMAs of the following quarters by a compile-time table (PLACE) — never a vmcnt(0) in the loop:
//      wait at the barrier of stage t, in flight oldest first: A(t+1), B(t+1), A(t+2) -> vmcnt(4) = A(t+2) may remain.
//  * Out-of-range rows (M, N not multiples of 256) and exhausted operand streams are zero-filled by the buffer
//    descriptor's bounds check (an exhausted stream's descriptor has length 0: no memory traffic).
//  * XCD-aware bijective workgroup -> tile map (8-tall column-major groups per XCD: the 32 workgroups an XCD runs at
//    once share their A / B panels in its L2).
//  * Epilogue through LDS (the ring is dead by then): XOR-swizzled park per wave, full 128-byte lines out (bias,
//    accumulate-into-C, optional transposed copy).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace tn {
namespace gemm {

constexpr int BM = 256, BN = 256;
constexpr int SLOT = 32768;                          // one operand stage: 256 rows x 64 k bf16
constexpr int LDS_BYTES = 5 * SLOT;
constexpr int NT = 512;
constexpr int MAXSEG = 3;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct Seg {
  const bf16_t* A;
  const bf16_t* B;
  long long lda, ldb;
  int K;
  int pad_;
};

struct Grp {
  bf16_t* C;
  long long ldc;
  int M, N, nbm, nbn;
  int start;           // index of the product's first tile in the launch's tile list (the running sum of nbm * nbn)
  int stages;          // ceil(K / 64)
};

// What the epilogue does with a finished tile:
//   EPI_PLAIN       C = acc (+ bias) (+ C), optional transposed copy
//   EPI_GROUPED     the same per product of a grouped launch (no bias / transposed copy)
//   EPI_SWIGLU_FWD  the MLP's gate and up products as ONE launch: an output tile is 256 rows x (128 gate columns + 128 up
//                   columns), the B stage image interleaves 32-row runs of gate_proj and up_proj so that every wave holds
//                   gate (block j = 0) and up (j = 1) of the SAME 32 columns in the same lanes; it writes gate, up (the
//                   backward needs both) and act = silu(gate) * up — the separate SwiGLU pass (3 tensors through HBM) is gone
//   EPI_SWIGLU_BWD  the product is d(act) = dY W_down; the epilogue reads the gate / up tiles through the park and writes
//                   d(gate), d(up): d(act) never exists in HBM and the separate backward pass (5 tensors) is gone
// (transformers' LlamaMLP.forward behind touchnet/models/llama/__init__.py:11-15, where the reference swaps in liger's
//  fused SwiGLU)
//   EPI_BIASG       EPI_PLAIN for a weight gradient dW = dY^T x that ALSO returns the bias gradient: the column sums of
//                   dY (the A operand, contraction-major) are taken from the A fragments the MFMAs read anyway — wave
//                   (wr, wc) of the tiles in output column 0 sums its A block wc over the whole contraction (16 VALU
//                   adds per 8 MFMAs) — so the separate column-sum pass over dY (one more trip of dY through HBM per
//                   biased layer: 258 launches per Qwen2-Audio step) is gone
//   EPI_ROPE        a q / k projection whose epilogue applies the rotary embedding (transformers' apply_rotary_pos_emb,
//                   modeling_llama.py:113-160 behind LlamaAttention.forward): the rotation pairs column c with c + D/2 of a
//                   head; for D = 128 the B stage image is arranged per DMA wave so that a wave's two 32-column blocks
//                   are (c .. c + 31) and (c + 64 .. c + 95) of one head — both members of every pair in ONE lane (D = 64:
//                   the natural layout already has that).  (acc + bias) is rounded to bf16 first, as the separate
//                   projection leaves it, then rotated with the row kernel's arithmetic: bit-identical, one pass over
//                   q and k through HBM less per layer
//   EPI_GELU_FWD    (round 6, the audio tower's fc1) C = x W^T + bias AND C2 = gelu(C): the activation pass read C again
//   EPI_GELU_BWD    (the tower's fc2 input gradient) the product d(act) = dY W never reaches HBM: C = d(act) o gelu'(E1),
//                   E1 = the saved fc1 output (pitch lde) — the GELU backward pass read both and wrote C
constexpr int EPI_PLAIN = 0, EPI_GROUPED = 1, EPI_SWIGLU_FWD = 2, EPI_SWIGLU_BWD = 3, EPI_BIASG = 4, EPI_ROPE = 5;
constexpr int EPI_GELU_FWD = 6, EPI_GELU_BWD = 7;

struct Params {
  Seg seg[MAXSEG];
  int nseg;
  int M, N;
  bf16_t* C;
  bf16_t* Ct;          // optional transposed copy [N, M] (ldct), or null
  const bf16_t* bias;  // optional [N]
  long long ldc, ldct;
  int accumulate;      // C += result
  // C = result + ADDEND (round 6: the residual stream added in the o_proj / down_proj epilogue — `hidden_states = residual
  // + hidden_states` of the HF decoder layers — so that the norm behind it reads ONE tensor): same arithmetic as
  // accumulate (the product rounded to bf16, added in fp32, rounded), the other term read from here instead of from C
  const bf16_t* addend;
  long long ldadd;
  int nbm, nbn;
  int stages;          // sum over segments of K / 64
  // split-K (outputs of few tiles with a deep contraction: the tower's 1280 x 1280 weight gradients over 30000 frames are
  // 25 tiles for 256 CUs): unit u = (split u / tiles, tile u % tiles) contracts stages [split * kchunk, (split+1) * kchunk)
  // of the ONE segment and leaves fp32 partial sums in ws[split][M][N]; splitk_reduce_kernel adds them up
  int splitk, kchunk;
  float* ws;           // [splitk][ntiles][256 x 256] fp32, tile-local
  // the linear tile ids (XCD-aware order, tile_of_block) this launch covers: all of them, or — "tail split" — the whole
  // rounds [0, main) unsplit in one launch and the last partial round [main, tiles) split-K in a second one (688 tiles on
  // 256 CUs are 2 rounds + 176 tiles: whole, the 176 take a third round at 69 % occupancy; cut in 4 they take 0.75)
  int tile0, ntiles;
  // fp32 output (weight gradients written straight into the data-parallel engine's fp32 staging buffers, utils/zero_dp.py):
  // C is a float matrix (ldc in floats), no bias, no transposed copy; accumulate adds to what is there
  int c_f32;
  // EPI_GROUPED: several independent products of one operand mode in ONE persistent launch — product g = seg[g] (one
  // segment each) with its own output; the launch walks the concatenation of their tile lists (tile0 / ntiles index it)
  Grp grp[MAXSEG];
  int ngrp;
  // EPI_SWIGLU_FWD (B = rows of gate_proj AND up_proj, seg[0].B / seg[1].B): C = gate, C2 = up, C3 = silu(gate) * up
  // EPI_SWIGLU_BWD (product = d(act)): E1 = gate, E2 = up (row pitch lde) are read, C = d(gate), C2 = d(up) are written
  bf16_t* C2;
  bf16_t* C3;
  const bf16_t* E1;
  const bf16_t* E2;
  long long lde;
  // EPI_BIASG: bias_out[M] (bf16) = column sums of A; under split-K the units leave fp32 partial sums in
  // bias_ws[split][nbm * 256] and the reduce kernel adds them up
  bf16_t* bias_out;
  float* bias_ws;
  // EPI_ROPE: cos / sin tables [M, rope_d / 2] (bf16, one row per output row), head dimension 64 or 128
  const bf16_t* rope_cos;
  const bf16_t* rope_sin;
  int rope_d;
};

// XCD-aware, bijective workgroup -> tile map
__device__ __forceinline__ void tile_of_block(int bid, int nbm, int nbn, int& tm, int& tn) {
  const int total = nbm * nbn;
  const int xcd = bid & 7, local = bid >> 3;
  const int q = total >> 3, r = total & 7;
  const int vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  constexpr int GM = 8;
  const int per_group = GM * nbn;
  const int g = vid / per_group, w = vid - g * per_group;
  const int gm = min(GM, nbm - g * GM);  // rows in this (possibly short, last) group
  tm = g * GM + w % gm;
  tn = w / gm;
}

typedef f32x16_t Acc[4][2];

// Timing experiments (scripts/build_gemm_variant.sh <name> -DTN_GEMM_ABLATE=n; results are garbage for n != 0):
//   1 no DMA in the loop   2 no fragment reads in the loop   3 no MFMA   4 no epilogue   5 no barrier in the loop
//   6 every DMA piece reads the tile's first rows (L1 / L2 hits only)   7 no vmcnt wait in front of the barrier (4-wave kernel)
#ifndef TN_GEMM_ABLATE
#define TN_GEMM_ABLATE 0
#endif

// Where the 8 DMA pieces a wave issues per stage go, as positions 0..31 in the stream of MFMAs that follows the barrier
// which freed their slots (0-7 = last quarter of stage t, 8-31 = quarters 0-2 of stage t+1).  All B positions must be
// below all A positions (the vmcnt arithmetic above relies on B(t+2) being older than A(t+3)).
//   0  burst right behind the barrier (the round-2 kernel)      1  B every other MFMA, A spread over the next two quarters
//   2  everything every fourth MFMA                              3  B in the last quarter, A one per following quarter pair
template <int PLACE> struct Place;
template <> struct Place<0> { static constexpr int B[4] = {0, 1, 2, 3}, A[4] = {4, 5, 6, 7}; };
template <> struct Place<1> { static constexpr int B[4] = {0, 2, 4, 6}, A[4] = {9, 13, 17, 21}; };
template <> struct Place<2> { static constexpr int B[4] = {0, 4, 8, 12}, A[4] = {16, 20, 24, 28}; };
template <> struct Place<3> { static constexpr int B[4] = {0, 2, 4, 6}, A[4] = {8, 12, 16, 20}; };
template <> struct Place<4> { static constexpr int B[4] = {0, 2, 4, 6}, A[4] = {8, 10, 12, 14}; };
template <> struct Place<5> { static constexpr int B[4] = {0, 1, 2, 3}, A[4] = {9, 13, 17, 21}; };
template <> struct Place<6> { static constexpr int B[4] = {0, 2, 4, 6}, A[4] = {14, 18, 22, 26}; };
template <> struct Place<7> { static constexpr int B[4] = {0, 2, 4, 6}, A[4] = {18, 22, 26, 30}; };   // (round 5, Kernel16 sweep)
template <> struct Place<8> { static constexpr int B[4] = {1, 3, 5, 7}, A[4] = {14, 18, 22, 26}; };
template <> struct Place<9> { static constexpr int B[4] = {0, 2, 4, 6}, A[4] = {16, 20, 24, 28}; };
// (Kernel16: a position = two MFMAs; a quarter's fragment reads sit behind its MFMAs 0..7 = its positions 0..3 — tables 10, 11
//  put every piece into the read-free second half of a quarter)
template <> struct Place<10> { static constexpr int B[4] = {4, 5, 6, 7}, A[4] = {12, 14, 20, 22}; };
template <> struct Place<11> { static constexpr int B[4] = {4, 5, 6, 7}, A[4] = {13, 15, 21, 23}; };

// piece index issued at position `pos` for the given table, or -1
template <int PLACE, bool IS_A> constexpr int piece_at(int pos) {
  for (int i = 0; i < 4; ++i)
    if ((IS_A ? Place<PLACE>::A[i] : Place<PLACE>::B[i]) == pos) return i;
  return -1;
}

// ---- operand stream: which bytes the next operand stage comes from -------------------------------------------------------
// (M16: the stage image the 16x16x32 kernel reads — contraction-major images swap the two 32-byte halves of every 64-byte
//  group in the k-rows with bit 3 set, see Kernel16)
template <bool KMAJ, bool M16 = false>
struct Stream {
  __amdgpu_buffer_rsrc_t rs;
  int soff;      // byte offset of the next stage inside the descriptor
  int step;      // bytes per stage
  int left;      // stages left in the current segment
  int seg;
  int ld2;       // row pitch in bytes of the current segment
  int voff[4];   // per-lane byte offsets of this wave's 4 pieces

  // (re)derive the per-lane piece offsets from the lane id (also used to re-materialise them behind a tile's epilogue,
  // so that they are not kept alive — i.e. spilled — across it)
  __device__ __forceinline__ void set_voff(int wave, int lane) {
    if constexpr (!KMAJ) {
      // piece q = rows 32 wave + 8 q + (lane >> 3), the lane's 16-byte slot holds chunk (lane & 7) ^ ((row >> 1) & 7)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = wave * 32 + q * 8 + (lane >> 3);
        voff[q] = (TN_GEMM_ABLATE == 6 ? (row & 7) : row) * ld2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
      }
    } else {
      // piece q = k-rows 8 wave + 2 q + (lane >> 5) of the stage; the lane's 16-byte slot u = lane & 31 of the 512-byte
      // row holds 64-byte group (u >> 2) ^ (k & 3), chunk u & 3
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = wave * 8 + q * 2 + (lane >> 5);
        const int u = lane & 31;
        const int ch = M16 ? ((u & 3) ^ (((k >> 3) & 1) << 1)) : (u & 3);
        voff[q] = (TN_GEMM_ABLATE == 6 ? (k & 1) : k) * ld2 + ((((u >> 2) ^ (k & 3)) << 6) | (ch << 4));
      }
    }
  }

  // R = rows of the output dimension this operand spans (M or N), origin = first one of this tile
  __device__ __forceinline__ void open(const bf16_t* X, long long ld, int K, int R, int origin, int wave, int lane) {
    ld2 = (int)(ld * 2);
    if constexpr (!KMAJ) {
      const long long bytes = max((long long)(R - origin) * ld2, 0LL);     // (origin >= R: nothing to read, all zeros)
      rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (long long)origin * ld), 0, (int)min(bytes, 0x7fffffffLL),
                                             0x00020000);
      step = 128;
    } else {
      const long long bytes = ((long long)(K - 1) * ld + (R - origin)) * 2;
      rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + origin), 0, (int)min(bytes, 0x7fffffffLL), 0x00020000);
      step = 64 * ld2;
    }
    set_voff(wave, lane);
    soff = 0;
    left = (K + 63) >> 6;
  }
  __device__ __forceinline__ void kill(const void* any) {
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)any, 0, 0, 0x00020000);   // length 0: every piece reads zeros, no traffic
    soff = 0;
    step = 0;
    left = 0x7fffffff;
  }
};

// MFMA operand fragments of one 16-deep quarter.  ROW fragments are plain LDS loads the compiler counts itself; KMAJ
// fragments are inline-asm transpose reads (the builtin carries no alias metadata: with an LDS-DMA pending hipcc puts
// `s_waitcnt vmcnt(0)` in front of it, which would serialise the ring) that are retired by wait_frags() below.
template <bool KMAJ, int NB>
struct Frags {
  bf16x8_t v[NB];
  u32x2_t h[KMAJ ? NB : 1][2];
};

template <int OFF>
__device__ __forceinline__ u32x2_t ds_tr16(uint32_t addr) {
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <int OFF>
__device__ __forceinline__ u32x4_t ds_b128(uint32_t addr) {
  u32x4_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <bool AK, bool BK, int PLACE, int ASYM, int ILV, bool HAS_CT, bool SPLITK = false, bool OUT_F32 = false,
          int EPI = EPI_PLAIN>
struct Kernel {
  static_assert(EPI == EPI_PLAIN || EPI == EPI_BIASG || (!HAS_CT && !SPLITK), "fused epilogues: whole tiles, no transposed copy");
  static_assert(EPI != EPI_BIASG || (AK && BK && !HAS_CT), "bias gradient: weight-gradient mode");
  static_assert(EPI != EPI_ROPE || (!AK && !BK && !OUT_F32), "RoPE epilogue: x W^T layout");
  static_assert(EPI != EPI_SWIGLU_FWD || (!AK && !BK && !OUT_F32), "SwiGLU forward: x W^T layout");
  static_assert(EPI != EPI_SWIGLU_BWD || (!AK && BK && !OUT_F32), "SwiGLU backward: dY W layout");
  static constexpr int BN_EFF = EPI == EPI_SWIGLU_FWD ? 128 : BN;   // output columns per tile (per matrix)
  // DMA wave w of the SwiGLU forward fetches rows [32 (w >> 1), + 32) of its matrix's 128-row panel: per-lane offsets as
  // for wave 0 (the bank swizzle only involves row bits below 32)
  struct Out {
    bf16_t* C;
    long long ldc;
    int M, N;
  };
  // ---- per-lane LDS read offsets ------------------------------------------------------------------------------------------
  //  ROW : xr[q] = (row0 + l31) * 128 + (((2 q + hi) ^ ((l31 >> 1) & 7)) << 4); block b at + b * 4096
  //  KMAJ: xk[b] = (8 hi + j) * 512 + half * 32 + w * 8 + (((blk0 + b) ^ j) << 6); quarter q, half e at + (16 q + 4 e) * 512
  //        (s4 = lane & 15, j = s4 >> 2 = k-row inside the 4-row group, w = s4 & 3, half = (lane >> 4) & 1)
  template <bool KMAJ, int NB>
  struct Reader {
    int x[4];
    __device__ __forceinline__ Reader(int lane, int row0 /* first of the wave's rows inside the 256-row tile */) {
      const int l31 = lane & 31, hi = lane >> 5;
      if constexpr (!KMAJ) {
        const int f = (l31 >> 1) & 7;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = (row0 + l31) * 128 + (((2 * q + hi) ^ f) << 4);
      } else {
        const int s4 = lane & 15, j = s4 >> 2, w = s4 & 3, half = (lane >> 4) & 1;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          x[b] = (8 * hi + j) * 512 + half * 32 + w * 8 + ((((row0 >> 5) + (b < NB ? b : 0)) ^ j) << 6);
      }
    }
    // fragment `b` of quarter Q from the slot at byte offset `sbase`
    template <int Q, int B>
    __device__ __forceinline__ void read(const char* smem, int sbase, Frags<KMAJ, NB>& f) const {
      if constexpr (!KMAJ) {
        f.v[B] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(smem + sbase + x[Q] + B * 4096));
      } else {
        const uint32_t a = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)(sbase + x[B]);
        f.h[B][0] = ds_tr16<(16 * Q) * 512>(a);
        f.h[B][1] = ds_tr16<(16 * Q + 4) * 512>(a);
      }
    }
  };

  static __device__ __forceinline__ void wait_frags(Frags<AK, 4>& a, Frags<BK, 2>& b) {
    if constexpr (AK && BK) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(a.h[0][0]), "+v"(a.h[0][1]), "+v"(a.h[1][0]), "+v"(a.h[1][1]), "+v"(a.h[2][0]), "+v"(a.h[2][1]),
                     "+v"(a.h[3][0]), "+v"(a.h[3][1]), "+v"(b.h[0][0]), "+v"(b.h[0][1]), "+v"(b.h[1][0]), "+v"(b.h[1][1]));
    } else if constexpr (AK) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(a.h[0][0]), "+v"(a.h[0][1]), "+v"(a.h[1][0]), "+v"(a.h[1][1]), "+v"(a.h[2][0]), "+v"(a.h[2][1]),
                     "+v"(a.h[3][0]), "+v"(a.h[3][1]));
    } else if constexpr (BK) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.h[0][0]), "+v"(b.h[0][1]), "+v"(b.h[1][0]), "+v"(b.h[1][1]));
    }
    if constexpr (AK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4_t t = {a.h[i][0].x, a.h[i][0].y, a.h[i][1].x, a.h[i][1].y};
        a.v[i] = __builtin_bit_cast(bf16x8_t, t);
      }
    }
    if constexpr (BK) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4_t t = {b.h[j][0].x, b.h[j][0].y, b.h[j][1].x, b.h[j][1].y};
        b.v[j] = __builtin_bit_cast(bf16x8_t, t);
      }
    }
  }

  // ---- the kernel body; HALF = 0 for waves 0-3, 1 for waves 4-7 (ASYM shifts the younger half's DMA positions by one
  //      MFMA so that the two waves of a SIMD, which leave every barrier together, do not issue their pieces in the same gap)
  template <int HALF>
  static __device__ __forceinline__ void body(const Params& p, char* smem) {
    constexpr int SH = ASYM ? HALF : 0;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;   // 2 x 4 waves

    // persistent: workgroup b of G owns tiles b, b + G, b + 2 G, ... of the XCD-aware order (G is a multiple of 8 or
    // the grid covers every tile at once, so a workgroup's tiles stay on its XCD)
    const int bid = blockIdx.x, G = gridDim.x;
    const int tiles = p.ntiles;
    const int mine = ((SPLITK ? tiles * p.splitk : tiles) - bid + G - 1) / G;
    auto origin_g = [&](int k, int& m0, int& n0, int& g) {
      int tm, tn, u = bid + k * G;
      if constexpr (SPLITK) u -= (u / tiles) * tiles;
      u += p.tile0;
      g = 0;
      if constexpr (EPI == EPI_GROUPED) {
        if (p.ngrp > 1 && u >= p.grp[1].start) g = 1;
        if (p.ngrp > 2 && u >= p.grp[2].start) g = 2;
        tile_of_block(u - p.grp[g].start, p.grp[g].nbm, p.grp[g].nbn, tm, tn);
      } else {
        tile_of_block(u, p.nbm, p.nbn, tm, tn);
      }
      m0 = tm * BM;
      n0 = tn * BN_EFF;
    };
    auto origin = [&](int k, int& m0, int& n0) {
      int g;
      origin_g(k, m0, n0, g);
    };
    auto split_of = [&](int k) { return (bid + k * G) / tiles; };

    Stream<AK> sA;
    Stream<BK> sB;
    int ka = 0, kb = 0;                        // tile cursors of the two operand streams (they run 2-3 stages ahead)
    auto open_a = [&](int s) {
      int m0, n0;
      origin(ka, m0, n0);
      if constexpr (SPLITK) {
        // the unit's share of the contraction: `kchunk` stages from k0 on (a share that starts behind K reads nothing;
        // one that crosses K is cut by the descriptor — contraction-major operands only, the host sees to that)
        const int k0 = split_of(ka) * p.kchunk * 64, klen = min(p.seg[0].K - k0, p.kchunk * 64);
        if (klen > 0) sA.open(p.seg[0].A + (AK ? (long long)k0 * p.seg[0].lda : (long long)k0), p.seg[0].lda, klen, p.M,
                              m0, wave, lane);
        else sA.kill(p.C);
        sA.left = p.kchunk;
      } else if constexpr (EPI == EPI_GROUPED) {
        int g;
        origin_g(ka, m0, n0, g);
        sA.open(p.seg[g].A, p.seg[g].lda, p.seg[g].K, p.grp[g].M, m0, wave, lane);
      } else {
        sA.open(p.seg[s].A, p.seg[s].lda, p.seg[s].K, p.M, m0, wave, lane);
      }
      sA.seg = s;
    };
    auto open_b = [&](int s) {
      int m0, n0;
      origin(kb, m0, n0);
      if constexpr (SPLITK) {
        const int k0 = split_of(kb) * p.kchunk * 64, klen = min(p.seg[0].K - k0, p.kchunk * 64);
        if (klen > 0) sB.open(p.seg[0].B + (BK ? (long long)k0 * p.seg[0].ldb : (long long)k0), p.seg[0].ldb, klen, p.N,
                              n0, wave, lane);
        else sB.kill(p.C);
        sB.left = p.kchunk;
      } else if constexpr (EPI == EPI_GROUPED) {
        int g;
        origin_g(kb, m0, n0, g);
        sB.open(p.seg[g].B, p.seg[g].ldb, p.seg[g].K, p.grp[g].N, n0, wave, lane);
      } else if constexpr (EPI == EPI_SWIGLU_FWD) {
        // LDS rows [32 w, 32 w + 32) of the B image = rows n0 + 32 (w >> 1) .. of gate_proj (w even) / up_proj (w odd)
        sB.open((wave & 1) ? p.seg[1].B : p.seg[0].B, p.seg[0].ldb, p.seg[0].K, p.N, n0 + (wave >> 1) * 32, 0, lane);
      } else if constexpr (EPI == EPI_ROPE) {
        // D = 128: LDS rows [32 w, 32 w + 32) = W rows of head n0 / 128 + (w >> 2), columns 64 (w & 1) + 32 ((w >> 1) & 1) ..:
        // reader wave wc = w >> 1 then holds (c, c + 64) pairs in its blocks j = 0 / 1.  D = 64: the plain order.
        if (p.rope_d == 128)
          sB.open(p.seg[0].B, p.seg[0].ldb, p.seg[0].K, p.N, n0 + (wave >> 2) * 128 + (wave & 1) * 64 + ((wave >> 1) & 1) * 32,
                  0, lane);
        else
          sB.open(p.seg[0].B, p.seg[0].ldb, p.seg[0].K, p.N, n0, wave, lane);
      } else {
        sB.open(p.seg[s].B, p.seg[s].ldb, p.seg[s].K, p.N, n0, wave, lane);
      }
      sB.seg = s;
    };
    auto adv_a = [&]() {
      sA.soff += sA.step;
      if (--sA.left == 0) {
        if (sA.seg + 1 < p.nseg) open_a(sA.seg + 1);
        else if (++ka < mine) open_a(0);
        else sA.kill(p.C);
      }
    };
    auto adv_b = [&]() {
      sB.soff += sB.step;
      if (--sB.left == 0) {
        if (sB.seg + 1 < p.nseg) open_b(sB.seg + 1);
        else if (++kb < mine) open_b(0);
        else sB.kill(p.C);
      }
    };
    auto piece_a = [&](int slot, int q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sA.rs, (lds_ptr_t)(smem + slot * SLOT + wave * 4096 + q * 1024), 16,
                                               sA.voff[q], sA.soff, 0, 0);
    };
    auto piece_b = [&](int slot, int q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sB.rs, (lds_ptr_t)(smem + slot * SLOT + wave * 4096 + q * 1024), 16,
                                               sB.voff[q], sB.soff, 0, 0);
    };
    open_a(0);
    open_b(0);

    Reader<AK, 4> ra(lane, wr * 128);
    Reader<BK, 2> rb(lane, wc * 64);

    Acc acc;
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    Frags<AK, 4> ae, ao;
    Frags<BK, 2> be, bo;

#define TN_PIN() __builtin_amdgcn_sched_barrier(0)
    // fragment f (0, 1 = B blocks; 2..5 = A blocks) of quarter Q
    auto read_frag = [&](auto QC, auto FC, int sa, int sb, Frags<AK, 4>& a, Frags<BK, 2>& b) {
      constexpr int Q = decltype(QC)::value, F = decltype(FC)::value;
      if constexpr (F < 2) rb.template read<Q, F>(smem, sb * SLOT, b);
      else ra.template read<Q, F - 2>(smem, sa * SLOT, a);
    };
    auto read_all = [&](auto QC, int sa, int sb, Frags<AK, 4>& a, Frags<BK, 2>& b) {
      read_frag(QC, std::integral_constant<int, 0>{}, sa, sb, a, b);
      read_frag(QC, std::integral_constant<int, 1>{}, sa, sb, a, b);
      read_frag(QC, std::integral_constant<int, 2>{}, sa, sb, a, b);
      read_frag(QC, std::integral_constant<int, 3>{}, sa, sb, a, b);
      read_frag(QC, std::integral_constant<int, 4>{}, sa, sb, a, b);
      read_frag(QC, std::integral_constant<int, 5>{}, sa, sb, a, b);
    };
    auto mma = [&](const Frags<AK, 4>& a, const Frags<BK, 2>& b, int i, int j) {
#if TN_GEMM_ABLATE == 3
      asm volatile("" ::"v"(a.v[i]), "v"(b.v[j]));
#else
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.v[j], a.v[i], acc[i][j], 0, 0, 0);
#endif
    };

    // EPI_BIASG: this wave's share of the A panel's column sums — block wc of the four A fragments (32 rows of dY^T), the
    // 8 contraction slots of the lane, every quarter of every stage (rows and depths beyond the operand are zero-filled)
    float bsum = 0.f;
    bool bias_on = false;
    auto bias_accum = [&](const Frags<AK, 4>& a) {
      auto add = [&](const bf16x8_t& v) {
        const u32x4_t w = __builtin_bit_cast(u32x4_t, v);
        bsum += ((__uint_as_float(w.x << 16) + __uint_as_float(w.x & 0xffff0000u)) +
                 (__uint_as_float(w.y << 16) + __uint_as_float(w.y & 0xffff0000u))) +
                ((__uint_as_float(w.z << 16) + __uint_as_float(w.z & 0xffff0000u)) +
                 (__uint_as_float(w.w << 16) + __uint_as_float(w.w & 0xffff0000u)));
      };
      if (wc == 0) add(a.v[0]);
      else if (wc == 1) add(a.v[1]);
      else if (wc == 2) add(a.v[2]);
      else add(a.v[3]);
    };

    // One quarter: the 8 MFMAs of (ca, cb); the fragments of quarter NQ of slots (nsa, nsb) are read into (na, nb) either
    // all in front (ILV = 0) or one behind each of the first six MFMAs (ILV = 1); the DMA pieces the placement table
    // puts at positions P0 .. P0 + 7 go behind their MFMA (P0 < 0: none).  dst_b / dst_a = slots the open piece set fills.
    auto quarter = [&](auto NQC, auto P0C, const Frags<AK, 4>& ca, const Frags<BK, 2>& cb, Frags<AK, 4>& na,
                       Frags<BK, 2>& nb, int nsa, int nsb, int dst_b, int dst_a) {
      constexpr int P0 = decltype(P0C)::value;
      if constexpr (ILV == 0 && TN_GEMM_ABLATE != 2) {
        read_all(NQC, nsa, nsb, na, nb);
        TN_PIN();
      }
      if constexpr (EPI == EPI_BIASG) {
        if (bias_on) bias_accum(ca);
        TN_PIN();
      }
      auto step = [&](auto MC) {
        constexpr int m = decltype(MC)::value;
        mma(ca, cb, m >> 1, m & 1);
        TN_PIN();
        if constexpr (ILV == 1 && m < 6 && TN_GEMM_ABLATE != 2) {
          read_frag(NQC, std::integral_constant<int, m>{}, nsa, nsb, na, nb);
          TN_PIN();
        }
        constexpr int pb = P0 < 0 ? -1 : piece_at<PLACE, false>(P0 + m - SH);
        constexpr int pa = P0 < 0 ? -1 : piece_at<PLACE, true>(P0 + m - SH);
        if constexpr (pb >= 0) {
          if constexpr (TN_GEMM_ABLATE != 1) piece_b(dst_b, pb);
          if constexpr (pb == 3) adv_b();
          TN_PIN();
        }
        if constexpr (pa >= 0) {
          if constexpr (TN_GEMM_ABLATE != 1) piece_a(dst_a, pa);
          if constexpr (pa == 3) adv_a();
          TN_PIN();
        }
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
    };
    // the pieces of positions 0..7 in one burst (prologue; behind a tile's epilogue)
    auto early_pieces = [&](int dst_b, int dst_a) {
      auto one = [&](auto MC) {
        constexpr int m = decltype(MC)::value;
        constexpr int pb = piece_at<PLACE, false>(m - SH), pa = piece_at<PLACE, true>(m - SH);
        if constexpr (pb >= 0) {
          piece_b(dst_b, pb);
          if constexpr (pb == 3) adv_b();
        }
        if constexpr (pa >= 0) {
          piece_a(dst_a, pa);
          if constexpr (pa == 3) adv_a();
        }
      };
      one(std::integral_constant<int, 0>{});
      one(std::integral_constant<int, 1>{});
      one(std::integral_constant<int, 2>{});
      one(std::integral_constant<int, 3>{});
      one(std::integral_constant<int, 4>{});
      one(std::integral_constant<int, 5>{});
      one(std::integral_constant<int, 6>{});
      one(std::integral_constant<int, 7>{});
    };
    auto next = [](int s, int d) { s += d; return s >= 5 ? s - 5 : s; };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using NONE = std::integral_constant<int, -1>;

    // prologue: A0 B0 A1 -> slots 0..2, then the piece set "opened at the barrier of stage -1" = {B(1) -> slot 3,
    // A(2) -> slot 4}: its pieces with positions < 8 are issued here, the later ones by the first trip
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(0, q);
    adv_a();
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b(1, q);
    adv_b();
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(2, q);
    adv_a();
    early_pieces(3, 4);
    constexpr int kEarly = (Place<PLACE>::B[0] + SH < 8) + (Place<PLACE>::B[1] + SH < 8) + (Place<PLACE>::B[2] + SH < 8) +
                           (Place<PLACE>::B[3] + SH < 8) + (Place<PLACE>::A[0] + SH < 8) + (Place<PLACE>::A[1] + SH < 8) +
                           (Place<PLACE>::A[2] + SH < 8) + (Place<PLACE>::A[3] + SH < 8);
    // A(0), B(0) must have landed; everything issued behind them may stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + kEarly) : "memory");
    __builtin_amdgcn_s_barrier();
    int sa = 0, sb = 1;                         // slots of A(t), B(t)
    int pa = 4, pb = 3;                         // slots the piece set opened at the previous barrier fills (A part, B part)

    // One stage.  LAST = last stage of a tile: its final quarter neither opens the next piece set (the two slots it
    // frees first serve as the epilogue's park) nor reads the next stage's first fragments (nothing but the accumulators
    // is to stay alive across the epilogue).
    auto trip = [&](auto LASTC) {
      constexpr bool LAST = decltype(LASTC)::value;
      const int sa1 = next(sa, 2), sb1 = next(sb, 2);       // slots of stage t+1
      __builtin_amdgcn_s_setprio(1);
      // quarters 0..2 of stage t (positions 8..31 of the piece set opened at the previous barrier)
      quarter(I1{}, std::integral_constant<int, 8>{}, ae, be, ao, bo, sa, sb, pb, pa);
      wait_frags(ao, bo);
      TN_PIN();
      quarter(I2{}, std::integral_constant<int, 16>{}, ao, bo, ae, be, sa, sb, pb, pa);
      wait_frags(ae, be);
      TN_PIN();
      quarter(I3{}, std::integral_constant<int, 24>{}, ae, be, ao, bo, sa, sb, pb, pa);
      __builtin_amdgcn_s_setprio(0);
      wait_frags(ao, bo);                                 // (asm reads retired here; ROW reads by the next line)
      __builtin_amdgcn_s_waitcnt(0xc07f);                 // my reads of stage t are complete (quarter 3 is in registers)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // A(t+1), B(t+1) landed; A(t+2) may still be in flight
      if constexpr (TN_GEMM_ABLATE != 5) __builtin_amdgcn_s_barrier();   // stage t+1 visible; slots sa, sb free
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!LAST) {
        // last quarter of stage t: the new piece set {B(t+2) -> slot sa, A(t+3) -> slot sb} opens (positions 0..7)
        quarter(I0{}, I0{}, ao, bo, ae, be, sa1, sb1, sa, sb);
        __builtin_amdgcn_s_setprio(0);
        wait_frags(ae, be);
        TN_PIN();
      } else {
        if constexpr (EPI == EPI_BIASG) {
          if (bias_on) bias_accum(ao);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) mma(ao, bo, m >> 1, m & 1);
        __builtin_amdgcn_s_setprio(0);
      }
      pb = sa;
      pa = sb;
      sa = sa1;
      sb = sb1;
    };

    const int np_all = SPLITK ? p.kchunk : p.stages;
    for (int kc = 0; kc < mine; ++kc) {
      int np = np_all;
      if constexpr (EPI == EPI_BIASG) {
        int m0_, n0_, g_;
        origin_g(kc, m0_, n0_, g_);
        bias_on = n0_ == 0;                                // the tiles of output column 0 carry the bias gradient
        bsum = 0.f;
      }
      if constexpr (EPI == EPI_GROUPED) {                  // (the products of a group may differ in depth)
        int m0_, n0_, g_;
        origin_g(kc, m0_, n0_, g_);
        np = p.grp[g_].stages;
      }
      read_all(I0{}, sa, sb, ae, be);
      wait_frags(ae, be);
      TN_PIN();
      for (int t = 1; t < np; ++t) trip(std::false_type{});
      trip(std::true_type{});
      // Epilogue.  The streams are already inside the next tile (its stage 0 and A(1) are in the other three slots);
      // the two slots the last barrier freed (now pb / pa) are the park, then they take their deferred pieces.
      int m0, n0, grp;
      origin_g(kc, m0, n0, grp);
      if constexpr (EPI == EPI_BIASG) {
        if (bias_on) {
          // the lane pair (l, l + 32) holds the two halves of every 16-deep slice of row l
          const float tot = bsum + __shfl_xor(bsum, 32, 64);
          const int m = m0 + wr * 128 + wc * 32 + (lane & 31);
          if (lane < 32) {
            if constexpr (SPLITK) {
              p.bias_ws[(long long)split_of(kc) * (p.nbm * BM) + m] = tot;
            } else if (m < p.M) {
              p.bias_out[m] = f2bf(tot);
            }
          }
        }
      }
      if constexpr (SPLITK) {
        const int u = bid + kc * G, sp = u / tiles;
        epilogue_ws(p, acc, sp * tiles + (u - sp * tiles), wr * 128, wc * 64, lane);
      } else if constexpr (OUT_F32) {
        if constexpr (EPI == EPI_GROUPED) {
          const Out o = {p.grp[grp].C, p.grp[grp].ldc, p.grp[grp].M, p.grp[grp].N};
          epilogue_f32(p, o, acc, m0 + wr * 128, n0 + wc * 64, lane);
        } else {
          const Out o = {p.C, p.ldc, p.M, p.N};
          epilogue_f32(p, o, acc, m0 + wr * 128, n0 + wc * 64, lane);
        }
      } else if constexpr (EPI == EPI_SWIGLU_FWD) {
        epilogue_swiglu_fwd(p, acc, smem + (wave < 4 ? pb : pa) * SLOT + (wave & 3) * 8192, m0 + wr * 128, n0 + wc * 32,
                            lane);
      } else if constexpr (EPI == EPI_SWIGLU_BWD) {
        epilogue_swiglu_bwd(p, acc, smem + (wave < 4 ? pb : pa) * SLOT + (wave & 3) * 8192, m0 + wr * 128, n0 + wc * 64,
                            lane);
      } else if constexpr (EPI == EPI_ROPE) {
        const bool d128 = p.rope_d == 128;
        const int col_a = d128 ? n0 + (wc >> 1) * 128 + (wc & 1) * 32 : n0 + wc * 64;     // block j = 0; j = 1: + D / 2
        epilogue_rope(p, acc, smem + (wave < 4 ? pb : pa) * SLOT + (wave & 3) * 8192, m0 + wr * 128, col_a,
                      d128 ? (wc & 1) * 32 : 0, lane);
      } else if constexpr (EPI == EPI_GROUPED) {
        const Out o = {p.grp[grp].C, p.grp[grp].ldc, p.grp[grp].M, p.grp[grp].N};
        epilogue(p, o, acc, smem + (wave < 4 ? pb : pa) * SLOT + (wave & 3) * 8192, m0 + wr * 128, n0 + wc * 64, lane);
      } else if constexpr (TN_GEMM_ABLATE != 4) {
        const Out o = {p.C, p.ldc, p.M, p.N};
        epilogue(p, o, acc, smem + (wave < 4 ? pb : pa) * SLOT + (wave & 3) * 8192, m0 + wr * 128, n0 + wc * 64, lane);
      }
      zero_acc();
      __builtin_amdgcn_s_waitcnt(0xc07f);               // my park reads are done ...
      __builtin_amdgcn_s_barrier();                     // ... and everybody's: the slots may be refilled
      // lane constants are re-derived from an opaque copy of the lane id: the compiler must not carry (spill) the old
      // ones across the epilogue
      asm volatile("" : "+v"(lane));
      ra = Reader<AK, 4>(lane, wr * 128);
      rb = Reader<BK, 2>(lane, wc * 64);
      sA.set_voff(wave, lane);
      sB.set_voff((EPI == EPI_SWIGLU_FWD || (EPI == EPI_ROPE && p.rope_d == 128)) ? 0 : wave, lane);
      early_pieces(pb, pa);
    }
  }

  static __device__ __forceinline__ void run(const Params& p, char* smem) {
    if constexpr (ASYM) {
      if (__builtin_amdgcn_readfirstlane(threadIdx.x) < 256) body<0>(p, smem);
      else body<1>(p, smem);
    } else {
      body<0>(p, smem);
    }
  }

  // fp32 output: the accumulators as they are, 16 bytes per lane and register quad (see epilogue_ws), clipped to M x N
  static __device__ __forceinline__ void epilogue_f32(const Params& p, const Out& o, Acc& acc, int wm0, int wn0, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    float* C = reinterpret_cast<float*>(o.C);
    const bool add = p.accumulate != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = wm0 + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = wn0 + j * 32 + 8 * g + 4 * hi;
          if (m < o.M && n < o.N) {                          // N is a multiple of 8: the quad is inside or outside
            f32x4_t* dst = reinterpret_cast<f32x4_t*>(C + (long long)m * o.ldc + n);
            f32x4_t v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            if (add) v += *dst;
            *dst = v;
          }
        }
    }
  }

  // Split-K epilogue: the fp32 accumulators go to this unit's [256 x 256] slab of the workspace as they are (a lane holds 4
  // consecutive n of one m per register quad: 16-byte stores; the L2 merges the quads of a line before it is written back).
  // Slabs are whole tiles: no bounds checks here, the reduce kernel clips to M x N.
  static __device__ __forceinline__ void epilogue_ws(const Params& p, Acc& acc, int slab, int lm0, int ln0, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    float* base = p.ws + (long long)slab * (BM * BN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = lm0 + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = ln0 + j * 32 + 8 * g + 4 * hi;
          const f32x4_t v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4_t*>(base + m * BN + n) = v;
        }
    }
  }

  // ---- fused SwiGLU epilogues ------------------------------------------------------------------------------------------
  // Arithmetic = tn::swiglu_fwd_kernel / swiglu_bwd_kernel (csrc/norm_act.hip) on the bf16-ROUNDED products, as the
  // separate passes saw them: silu(gate) rounded to bf16 before the product (what the eager path materialises), so the
  // fused and the unfused MLP agree bit for bit.
  static __device__ __forceinline__ float sigm(float x) { return sigmoid_fast(x); }
  static __device__ __forceinline__ float lo16(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ float rbf(float v) { return __uint_as_float(((uint32_t)f2bf(v)) << 16); }

  // Forward.  The wave holds gate (acc[i][0]) and up (acc[i][1]) of rows wm0 .. wm0 + 127, columns wn0 .. wn0 + 31.
  // Three trips through its 8 KB park ([64 rows][64 columns] bf16, 128-byte rows, 16-byte chunk ^= row & 7):
  //   1, 2  rows half * 64 ..: columns 0-31 = gate, 32-63 = up  -> 64-byte row segments of C (gate) and C2 (up)
  //   3     act of ALL 128 rows: columns 0-31 = rows 0-63, columns 32-63 = rows 64-127 -> 64-byte row segments of C3
  static __device__ __forceinline__ void epilogue_swiglu_fwd(const Params& p, Acc& acc, char* park, int wm0, int wn0,
                                                             int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    // (the per-lane choice between the two outputs as integer arithmetic on pointers held in registers: written as
    //  `c < 4 ? p.C : p.C2` hipcc selects between the two kernel-argument ADDRESSES and re-loads the pointer per store)
    const uintptr_t c_gate = (uintptr_t)p.C, c_up = (uintptr_t)p.C2;
    bf16_t* const gu_base = reinterpret_cast<bf16_t*>(c_gate + ((lane & 4) ? c_up - c_gate : (uintptr_t)0));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = half * 2 + ii;
        const int row = ii * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int chunk = (j * 4 + g) ^ (row & 7);
            const u32x2_t pk = {pack2bf(acc[i][j][4 * g], acc[i][j][4 * g + 1]),
                                pack2bf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])};
            *reinterpret_cast<u32x2_t*>(park + row * 128 + chunk * 16 + hi * 8) = pk;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
        const int m = wm0 + half * 64 + row, n = wn0 + (c & 3) * 8;
        if (m < p.M && n < p.N)
          *reinterpret_cast<uint4*>(gu_base + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (i & 1) * 32 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gf = rbf(acc[i][0][4 * g + e]), uf = rbf(acc[i][1][4 * g + e]);
          h[e] = rbf(gf * sigm(gf)) * uf;
        }
        const int chunk = ((i >> 1) * 4 + g) ^ (row & 7);
        const u32x2_t pk = {pack2bf(h[0], h[1]), pack2bf(h[2], h[3])};
        *reinterpret_cast<u32x2_t*>(park + row * 128 + chunk * 16 + hi * 8) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
      const int m = wm0 + (c >> 2) * 64 + row, n = wn0 + (c & 3) * 8;
      if (m < p.M && n < p.N)
        *reinterpret_cast<uint4*>(p.C3 + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
    }
  }

  // RoPE.  The wave holds columns col_a .. + 31 (acc[i][0]) and col_a + D/2 .. + 31 (acc[i][1]) of rows wm0 .. + 127: both
  // members of every rotary pair in one lane; c0 = the first pair's index inside the head (table column).
  static __device__ __forceinline__ void epilogue_rope(const Params& p, Acc& acc, char* park, int wm0, int col_a, int c0,
                                                       int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    const int half = p.rope_d >> 1;
    const int col_b = col_a + half;
    // every table / bias element of the tile is requested up front (32 + 8 loads in flight per lane, ONE wait per 64-row
    // half) — issued one (row block, column quad) at a time the epilogue paid a memory round trip per quad
    uint2 bwa[4], bwb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q = 8 * g + 4 * hi;
      bwa[g] = bwb[g] = make_uint2(0, 0);
      if (p.bias != nullptr) {
        bwa[g] = *reinterpret_cast<const uint2*>(p.bias + min(col_a + q, p.N - 4));
        bwb[g] = *reinterpret_cast<const uint2*>(p.bias + min(col_b + q, p.N - 4));
      }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      uint2 cw[2][4], sw[2][4];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int m = min(wm0 + (hf * 2 + ii) * 32 + l31, p.M - 1);         // (rows >= M are never stored)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long long o = (long long)m * half + c0 + 8 * g + 4 * hi;
          cw[ii][g] = *reinterpret_cast<const uint2*>(p.rope_cos + o);
          sw[ii][g] = *reinterpret_cast<const uint2*>(p.rope_sin + o);
        }
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = hf * 2 + ii;
        const int row = ii * 32 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float ba[4] = {lo16(bwa[g].x), hi16(bwa[g].x), lo16(bwa[g].y), hi16(bwa[g].y)};
          const float bb[4] = {lo16(bwb[g].x), hi16(bwb[g].x), lo16(bwb[g].y), hi16(bwb[g].y)};
          const float cf[4] = {lo16(cw[ii][g].x), hi16(cw[ii][g].x), lo16(cw[ii][g].y), hi16(cw[ii][g].y)};
          const float sf[4] = {lo16(sw[ii][g].x), hi16(sw[ii][g].x), lo16(sw[ii][g].y), hi16(sw[ii][g].y)};
          float ya[4], yb[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            rope_rotate(rbf(acc[i][0][4 * g + e] + ba[e]), rbf(acc[i][1][4 * g + e] + bb[e]), cf[e], sf[e], ya[e], yb[e]);
          const u32x2_t pa_ = {pack2bf(ya[0], ya[1]), pack2bf(ya[2], ya[3])};
          const u32x2_t pb_ = {pack2bf(yb[0], yb[1]), pack2bf(yb[2], yb[3])};
          *reinterpret_cast<u32x2_t*>(park + row * 128 + ((g ^ (row & 7)) << 4) + hi * 8) = pa_;
          *reinterpret_cast<u32x2_t*>(park + row * 128 + (((4 + g) ^ (row & 7)) << 4) + hi * 8) = pb_;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
        const int m = wm0 + hf * 64 + row, n = (c < 4 ? col_a : col_b) + (c & 3) * 8;
        if (m < p.M && n < p.N)
          *reinterpret_cast<uint4*>(p.C + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
      }
    }
  }

  // Backward.  acc = d(act) of the wave's 128 x 64 tile.  Per 64-row half: the gate and up rows come in as full 128-byte
  // lines (16 bytes per lane), go through the park into the accumulator layout (row = lane, 4 consecutive columns per
  // register quad), d(gate) / d(up) are formed in registers and leave through the park like any other tile.
  static __device__ __forceinline__ void epilogue_swiglu_bwd(const Params& p, Acc& acc, char* park, int wm0, int wn0,
                                                             int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 gv[8], uv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const int m = wm0 + half * 64 + row, n = wn0 + c * 8;
        const bool in = m < p.M && n < p.N;
        const long long off = (long long)m * p.lde + n;
        gv[it] = in ? *reinterpret_cast<const uint4*>(p.E1 + off) : make_uint4(0, 0, 0, 0);
        uv[it] = in ? *reinterpret_cast<const uint4*>(p.E2 + off) : make_uint4(0, 0, 0, 0);
      }
      u32x2_t gq[2][2][4], uq[2][2][4];
      auto through_park = [&](const uint4 (&src)[8], u32x2_t (&dst)[2][2][4]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), c = lane & 7;
          const u32x4_t v = {src[it].x, src[it].y, src[it].z, src[it].w};
          *reinterpret_cast<u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int row = ii * 32 + l31;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              dst[ii][j][g] =
                  *reinterpret_cast<const u32x2_t*>(park + row * 128 + (((j * 4 + g) ^ (row & 7)) << 4) + hi * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      through_park(gv, gq);
      through_park(uv, uq);
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = half * 2 + ii;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float dg[4], du[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t wg = e < 2 ? gq[ii][j][g].x : gq[ii][j][g].y, wu = e < 2 ? uq[ii][j][g].x : uq[ii][j][g].y;
              const float gf = (e & 1) ? hi16(wg) : lo16(wg), uf = (e & 1) ? hi16(wu) : lo16(wu);
              const float d = rbf(acc[i][j][4 * g + e]);
              const float sg = sigm(gf);
              const float silu = gf * sg;
              du[e] = d * silu;
              dg[e] = d * uf * (sg + silu * (1.f - sg));
            }
            gq[ii][j][g] = u32x2_t{pack2bf(dg[0], dg[1]), pack2bf(dg[2], dg[3])};
            uq[ii][j][g] = u32x2_t{pack2bf(du[0], du[1]), pack2bf(du[2], du[3])};
          }
      }
      auto out_park = [&](const u32x2_t (&src)[2][2][4], bf16_t* C) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int row = ii * 32 + l31;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<u32x2_t*>(park + row * 128 + (((j * 4 + g) ^ (row & 7)) << 4) + hi * 8) = src[ii][j][g];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), c = lane & 7;
          const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
          const int m = wm0 + half * 64 + row, n = wn0 + c * 8;
          if (m < p.M && n < p.N)
            *reinterpret_cast<uint4*>(C + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
        }
      };
      out_park(gq, p.C);
      out_park(uq, p.C2);
    }
  }

  // Epilogue through LDS: the wave parks its 128 x 64 tile, 64 rows at a time, in its own XOR-swizzled 8 KB
  // ([64 rows][64 cols] bf16, 128-byte rows, chunk ^= row & 7) and writes full 128-byte lines.
  static __device__ __forceinline__ void epilogue(const Params& p, const Out& o, Acc& acc, char* park, int wm0, int wn0,
                                                  int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    float bias_v[2][4][4];
    if (p.bias != nullptr) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = min(wn0 + j * 32 + 8 * g + 4 * hi, o.N - 4);    // (columns >= N are never stored)
          const uint2 w = *reinterpret_cast<const uint2*>(p.bias + n);
          bias_v[j][g][0] = __uint_as_float(w.x << 16);
          bias_v[j][g][1] = __uint_as_float(w.x & 0xffff0000u);
          bias_v[j][g][2] = __uint_as_float(w.y << 16);
          bias_v[j][g][3] = __uint_as_float(w.y & 0xffff0000u);
        }
    }
    const bool acc_c = p.accumulate != 0 || p.addend != nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // result rows (registers) = n, result column (lane) = m: 4 consecutive n per register quad -> 8-byte packs
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = half * 2 + ii;
        const int row = ii * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + (p.bias != nullptr ? bias_v[j][g][e] : 0.f);
            const int chunk = (j * 4 + g) ^ (row & 7);
            // (vector-typed LDS accesses: hipcc puts `s_waitcnt vmcnt(0)` in front of an LDS access without alias
            //  metadata while an LDS-DMA is pending — here the next tile's prefetched stages)
            const u32x2_t pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2_t*>(park + row * 128 + chunk * 16 + hi * 8) = pk;
          }
      }
      // (only this wave touches its park region: a wave-level wait is enough; LDS operations of one wave execute in
      //  order, so the next half's stores cannot overtake this half's reads)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
        uint4 v = make_uint4(pv.x, pv.y, pv.z, pv.w);
        const int m = wm0 + half * 64 + row, n = wn0 + c * 8;
        if (m < o.M && n < o.N) {                            // N is a multiple of 8 (checked by the host)
          bf16_t* dst = o.C + (long long)m * o.ldc + n;
          if (acc_c) {
            Vec16<bf16_t> o, nw;
            o.load(p.addend != nullptr ? p.addend + (long long)m * p.ldadd + n : dst);
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
        // transposed copy: Ct[n, m]; a lane gathers 8 consecutive m of one n from the parked half tile (2-byte LDS
        // reads: this path trades LDS instructions for the HBM round trip of a separate transpose pass)
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
          const int n_l = it * 8 + (lane >> 3), mg = lane & 7;       // 64 n x 8 groups of 8 m
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
          const int n = wn0 + n_l, m = wm0 + half * 64 + mg * 8;
          if (n < o.N && m < o.M)     // M is a multiple of 8 when a transposed copy is requested (host check)
            *reinterpret_cast<uint4*>(p.Ct + (long long)n * p.ldct + m) = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
  }
};

template <bool AK, bool BK, int PLACE, int ASYM, int ILV, bool HAS_CT, bool SPLITK = false, bool OUT_F32 = false,
          int EPI = EPI_PLAIN>
__global__ __launch_bounds__(NT, 2) void gemm_kernel(const Params p) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
  Kernel<AK, BK, PLACE, ASYM, ILV, HAS_CT, SPLITK, OUT_F32, EPI>::run(p, smem);
}

// =====================================================================================================================
// Kernel16: the same workgroup geometry, stage ring, DMA placement and persistent tile walk as Kernel above, with
// v_mfma_f32_16x16x32_bf16 instead of v_mfma_f32_32x32x16_bf16.  Why (round 5, profiles/r05p_*): with operands in
// registers only the 32x32x16 shape SUSTAINS 1.73-1.78 PF on this chip — the 1.67 GHz the GEMM was always measured at —
// and the 16x16x32 shape 1.98-1.99 PF (half the accumulator register traffic per flop): the "power limit" of the
// kernel was its instruction shape (hipBLASLt's gfx950 kernels issue 16x16x32).
//  * wave tile 128 x 64 = 8 x 4 blocks of 16 x 16, accumulators f32x4 [8][4] (the same 128 registers)
//  * a 64-deep stage = two 32-deep halves; a "quarter" = 16 MFMAs = A blocks 4 part .. 4 part + 3 (part = quarter & 1) of
//    half quarter >> 1 against the half's four B blocks.  Fragment sets: 4 A fragments per quarter (double buffered),
//    4 B fragments per half (double buffered); one fragment read behind every second MFMA.
//  * operand fragment of v_mfma_f32_16x16x32: lane l holds row l & 15, contraction slots 8 (l >> 4) .. + 7:
//      ROW   one ds_read_b128 at row * 128 + ((4 half + (l >> 4)) ^ ((row >> 1) & 7)) * 16   (same image as Kernel)
//      KMAJ  two ds_read_b64_tr_b16: the 16 lanes of group g4 = l >> 4 point at k-rows 32 half + 8 g4 + (0..3) [+ 4] x the
//            32 bytes of a 16-row block.  Lanes 0-15 and 16-31 are served together and would hit the same banks (their
//            k-rows are 8 apart: same (k & 3) group swizzle), so the IMAGE swaps the two 32-byte halves of every 64-byte
//            group in k-rows with bit 3 set (Stream<.., M16>): the second group reads the other half.
//  * result of mfma(b, a): lane holds column m = l & 15 of its 16 x 16 block, registers r = rows n = 4 (l >> 4) + r
// =====================================================================================================================
typedef f32x4_t Acc16[8][4];
#ifndef TN_G16_RSTEP
#define TN_G16_RSTEP 1
#endif

template <bool AK, bool BK, int PLACE, int EPI = EPI_PLAIN>
struct Kernel16 {
  static_assert(EPI == EPI_PLAIN || EPI == EPI_SWIGLU_FWD || EPI == EPI_SWIGLU_BWD || EPI == EPI_ROPE ||
                    EPI == EPI_GELU_FWD || EPI == EPI_GELU_BWD,
                "Kernel16: plain / SwiGLU / RoPE / GELU epilogues (the weight-gradient modes stay on Kernel)");
  static_assert(EPI != EPI_GELU_FWD || !BK, "GELU forward: x W^T layout");
  static_assert(EPI != EPI_GELU_BWD || BK, "GELU backward: dY W layout");
  static_assert(!AK, "Kernel16 is used where it is faster: row-stored A (forward and input-gradient products)");
  static_assert(EPI != EPI_SWIGLU_FWD || !BK, "SwiGLU forward: x W^T layout");
  static_assert(EPI != EPI_SWIGLU_BWD || BK, "SwiGLU backward: dY W layout");
  static_assert(EPI != EPI_ROPE || !BK, "RoPE epilogue: x W^T layout");
  static constexpr int BN_EFF = EPI == EPI_SWIGLU_FWD ? 128 : BN;
  static __device__ __forceinline__ float sigm(float x) { return sigmoid_fast(x); }
  static __device__ __forceinline__ float lo16(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi16(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ float rbf(float v) { return __uint_as_float(((uint32_t)f2bf(v)) << 16); }
  template <bool KMAJ>
  struct F4 {
    bf16x8_t v[4];
    u32x2_t h[KMAJ ? 4 : 1][2];
#ifdef TN_G16_ASMROW
    u32x4_t r[4];
#endif
  };
  // per-lane LDS offsets.  ROW: x[half]; block b of the wave's rows at + b * 2048.  KMAJ: x[G] for the 64-byte group G of
  // the wave's 32-row pairs, x ^ 32 for the odd 16-row block of the pair; k-row offsets are compile-time.
  template <bool KMAJ, int NB /* 16-row blocks: 8 (A) or 4 (B) */>
  struct Reader {
    int x[KMAJ ? NB / 2 : 2];
    __device__ __forceinline__ Reader(int lane, int row0) {
      const int l15 = lane & 15, g4 = lane >> 4;
      if constexpr (!KMAJ) {
        const int f = (l15 >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) x[h] = (row0 + l15) * 128 + (((4 * h + g4) ^ f) << 4);
      } else {
        const int j = l15 >> 2, w = l15 & 3;
#pragma unroll
        for (int g = 0; g < NB / 2; ++g)
          x[g] = (8 * g4 + j) * 512 + w * 8 + ((((row0 >> 5) + g) ^ j) << 6) + (g4 & 1) * 32;
      }
    }
    // fragment I (0..3) of the set: 16-row block BLK of the wave's rows, half H of the stage
    template <int H, int BLK, int I>
    __device__ __forceinline__ void read(const char* smem, int sbase, F4<KMAJ>& f) const {
      if constexpr (!KMAJ) {
#ifdef TN_G16_ASMROW   // diagnostic: row fragments as asm reads retired by ONE lgkmcnt(0) per set, like the transpose reads
        const uint32_t a = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)(sbase + x[H]);
        f.r[I] = ds_b128<BLK * 2048>(a);
#else
        f.v[I] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(smem + sbase + x[H] + BLK * 2048));
#endif
      } else {
        const uint32_t a = (uint32_t)(size_t)(lds_ptr_t)smem + (uint32_t)(sbase + (x[BLK >> 1] ^ ((BLK & 1) * 32)));
        f.h[I][0] = ds_tr16<(32 * H) * 512>(a);
        f.h[I][1] = ds_tr16<(32 * H + 4) * 512>(a);
      }
    }
  };
  template <bool KMAJ>
  static __device__ __forceinline__ void retire(F4<KMAJ>& f) {
#ifdef TN_G16_ASMROW
    if constexpr (!KMAJ) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.r[0]), "+v"(f.r[1]), "+v"(f.r[2]), "+v"(f.r[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i) f.v[i] = __builtin_bit_cast(bf16x8_t, f.r[i]);
    }
#endif
    if constexpr (KMAJ) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.h[0][0]), "+v"(f.h[0][1]), "+v"(f.h[1][0]), "+v"(f.h[1][1]), "+v"(f.h[2][0]), "+v"(f.h[2][1]),
                     "+v"(f.h[3][0]), "+v"(f.h[3][1]));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4_t t = {f.h[i][0].x, f.h[i][0].y, f.h[i][1].x, f.h[i][1].y};
        f.v[i] = __builtin_bit_cast(bf16x8_t, t);
      }
    }
  }

  static __device__ __forceinline__ void run(const Params& p, char* smem) {
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = blockIdx.x, G = gridDim.x;
    const int tiles = p.ntiles;
    const int mine = (tiles - bid + G - 1) / G;
    auto origin = [&](int k, int& m0, int& n0) {
      int tm, tn;
      tile_of_block(p.tile0 + bid + k * G, p.nbm, p.nbn, tm, tn);
      m0 = tm * BM;
      n0 = tn * BN_EFF;
    };
    Stream<AK, true> sA;
    Stream<BK, true> sB;
    int ka = 0, kb = 0;
    auto open_a = [&](int s) {
      int m0, n0;
      origin(ka, m0, n0);
      sA.open(p.seg[s].A, p.seg[s].lda, p.seg[s].K, p.M, m0, wave, lane);
      sA.seg = s;
    };
    // (B stage images of the fused epilogues: the per-DMA-wave row choices of Kernel::open_b — they concern LDS rows, not
    //  the MFMA shape)
    const bool b_wave0 = EPI == EPI_SWIGLU_FWD || (EPI == EPI_ROPE && p.rope_d == 128);
    auto open_b = [&](int s) {
      int m0, n0;
      origin(kb, m0, n0);
      if constexpr (EPI == EPI_SWIGLU_FWD) {
        sB.open((wave & 1) ? p.seg[1].B : p.seg[0].B, p.seg[0].ldb, p.seg[0].K, p.N, n0 + (wave >> 1) * 32, 0, lane);
      } else if constexpr (EPI == EPI_ROPE) {
        if (p.rope_d == 128)
          sB.open(p.seg[0].B, p.seg[0].ldb, p.seg[0].K, p.N, n0 + (wave >> 2) * 128 + (wave & 1) * 64 + ((wave >> 1) & 1) * 32,
                  0, lane);
        else
          sB.open(p.seg[0].B, p.seg[0].ldb, p.seg[0].K, p.N, n0, wave, lane);
      } else {
        sB.open(p.seg[s].B, p.seg[s].ldb, p.seg[s].K, p.N, n0, wave, lane);
      }
      sB.seg = s;
    };
    auto adv_a = [&]() {
      sA.soff += sA.step;
      if (--sA.left == 0) {
        if (sA.seg + 1 < p.nseg) open_a(sA.seg + 1);
        else if (++ka < mine) open_a(0);
        else sA.kill(p.C);
      }
    };
    auto adv_b = [&]() {
      sB.soff += sB.step;
      if (--sB.left == 0) {
        if (sB.seg + 1 < p.nseg) open_b(sB.seg + 1);
        else if (++kb < mine) open_b(0);
        else sB.kill(p.C);
      }
    };
    auto piece_a = [&](int slot, int q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sA.rs, (lds_ptr_t)(smem + slot * SLOT + wave * 4096 + q * 1024), 16,
                                               sA.voff[q], sA.soff, 0, 0);
    };
    auto piece_b = [&](int slot, int q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sB.rs, (lds_ptr_t)(smem + slot * SLOT + wave * 4096 + q * 1024), 16,
                                               sB.voff[q], sB.soff, 0, 0);
    };
    open_a(0);
    open_b(0);

    Reader<AK, 8> ra(lane, wr * 128);
    Reader<BK, 4> rb(lane, wc * 64);
    Acc16 acc;
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    F4<AK> ae, ao;
    F4<BK> be, bo;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    // read fragment R (0..3 = A blocks of quarter Q, 4..7 = the B blocks of half Q >> 1) of the quarter that follows
    auto read_frag = [&](auto QC, auto RC, int sa, int sb, F4<AK>& a, F4<BK>& b) {
      constexpr int Q = decltype(QC)::value, R = decltype(RC)::value;
      if constexpr (R < 4) ra.template read<(Q >> 1), (Q & 1) * 4 + R, R>(smem, sa * SLOT, a);
      else rb.template read<(Q >> 1), R - 4, R - 4>(smem, sb * SLOT, b);
    };
    // (Measured and dropped, round 5: B fragments read first and retired by a counted `lgkmcnt(4)` while the four younger A
    //  reads stay in flight — 0.7 % SLOWER than A first + lgkmcnt(0), profiles/r05s_*.)
    // One quarter: 16 MFMAs of A part PART (fragments ca) against the half's B fragments cb; fragment r of the NEXT quarter
    // NQ (4 A fragments, and with RB its half's 4 B fragments) is read behind MFMA 2 r; DMA pieces of positions
    // P0 .. P0 + 7 go behind MFMAs 0, 2, .. 14 (P0 < 0: none).
    auto quarter = [&](auto NQC, auto RBC, auto P0C, auto PARTC, const F4<AK>& ca, const F4<BK>& cb, F4<AK>& na,
                       F4<BK>& nb, int nsa, int nsb, int dst_b, int dst_a) {
      constexpr int P0 = decltype(P0C)::value, PART = decltype(PARTC)::value;
      constexpr bool RB = decltype(RBC)::value;
      auto step = [&](auto MC) {
        constexpr int m = decltype(MC)::value, i = m >> 2, j = m & 3;
        acc[PART * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb.v[j], ca.v[i], acc[PART * 4 + i][j], 0, 0, 0);
        TN_PIN();
        // fragment reads behind the FIRST MFMAs of the quarter (TN_G16_RSTEP = MFMAs per read: 1 = reads 0..7 behind
        // MFMAs 0..7, so the last one is 8+ MFMAs old when the quarter's fragments are retired)
        if constexpr ((m % TN_G16_RSTEP) == 0 && m / TN_G16_RSTEP < 8) {
          constexpr int r = m / TN_G16_RSTEP;
          if constexpr (r < 4 || RB) {
            read_frag(NQC, std::integral_constant<int, r>{}, nsa, nsb, na, nb);
            TN_PIN();
          }
        }
        if constexpr ((m & 1) == 0) {
          constexpr int r = m >> 1;
          constexpr int pb = P0 < 0 ? -1 : piece_at<PLACE, false>(P0 + r);
          constexpr int pa = P0 < 0 ? -1 : piece_at<PLACE, true>(P0 + r);
          if constexpr (pb >= 0) {
            piece_b(dst_b, pb);
            if constexpr (pb == 3) adv_b();
            TN_PIN();
          }
          if constexpr (pa >= 0) {
            piece_a(dst_a, pa);
            if constexpr (pa == 3) adv_a();
            TN_PIN();
          }
        }
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
      step(std::integral_constant<int, 8>{});
      step(std::integral_constant<int, 9>{});
      step(std::integral_constant<int, 10>{});
      step(std::integral_constant<int, 11>{});
      step(std::integral_constant<int, 12>{});
      step(std::integral_constant<int, 13>{});
      step(std::integral_constant<int, 14>{});
      step(std::integral_constant<int, 15>{});
    };
    auto early_pieces = [&](int dst_b, int dst_a) {
      auto one = [&](auto MC) {
        constexpr int m = decltype(MC)::value;
        constexpr int pb = piece_at<PLACE, false>(m), pa = piece_at<PLACE, true>(m);
        if constexpr (pb >= 0) {
          piece_b(dst_b, pb);
          if constexpr (pb == 3) adv_b();
        }
        if constexpr (pa >= 0) {
          piece_a(dst_a, pa);
          if constexpr (pa == 3) adv_a();
        }
      };
      one(std::integral_constant<int, 0>{});
      one(std::integral_constant<int, 1>{});
      one(std::integral_constant<int, 2>{});
      one(std::integral_constant<int, 3>{});
      one(std::integral_constant<int, 4>{});
      one(std::integral_constant<int, 5>{});
      one(std::integral_constant<int, 6>{});
      one(std::integral_constant<int, 7>{});
    };
    auto next = [](int s, int d) { s += d; return s >= 5 ? s - 5 : s; };
    using T_ = std::true_type;
    using F_ = std::false_type;

#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(0, q);
    adv_a();
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_b(1, q);
    adv_b();
#pragma unroll
    for (int q = 0; q < 4; ++q) piece_a(2, q);
    adv_a();
    early_pieces(3, 4);
    constexpr int kEarly = (Place<PLACE>::B[0] < 8) + (Place<PLACE>::B[1] < 8) + (Place<PLACE>::B[2] < 8) +
                           (Place<PLACE>::B[3] < 8) + (Place<PLACE>::A[0] < 8) + (Place<PLACE>::A[1] < 8) +
                           (Place<PLACE>::A[2] < 8) + (Place<PLACE>::A[3] < 8);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + kEarly) : "memory");
    __builtin_amdgcn_s_barrier();
    int sa = 0, sb = 1;
    int pa = 4, pb = 3;

    auto trip = [&](auto LASTC) {
      constexpr bool LAST = decltype(LASTC)::value;
      const int sa1 = next(sa, 2), sb1 = next(sb, 2);
      __builtin_amdgcn_s_setprio(1);
      // quarter 0 (half 0, A part 0): reads A of quarter 1
      quarter(I1{}, F_{}, std::integral_constant<int, 8>{}, I0{}, ae, be, ao, bo, sa, sb, pb, pa);
      retire(ao);
      TN_PIN();
      // quarter 1 (half 0, part 1): reads A of quarter 2 and B of half 1
      quarter(I2{}, T_{}, std::integral_constant<int, 16>{}, I1{}, ao, be, ae, bo, sa, sb, pb, pa);
      retire(ae);
      retire(bo);
      TN_PIN();
      // quarter 2 (half 1, part 0): reads A of quarter 3
      quarter(I3{}, F_{}, std::integral_constant<int, 24>{}, I0{}, ae, bo, ao, be, sa, sb, pb, pa);
      __builtin_amdgcn_s_setprio(0);
      retire(ao);
      __builtin_amdgcn_s_waitcnt(0xc07f);                 // my reads of stage t are complete
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // A(t+1), B(t+1) landed; A(t+2) may still be in flight
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!LAST) {
        // quarter 3 (half 1, part 1): the new piece set opens; reads A of quarter 0 and B of half 0 of stage t + 1
        quarter(I0{}, T_{}, I0{}, I1{}, ao, bo, ae, be, sa1, sb1, sa, sb);
        __builtin_amdgcn_s_setprio(0);
        retire(ae);
        retire(be);
        TN_PIN();
      } else {
#pragma unroll
        for (int m = 0; m < 16; ++m)
          acc[4 + (m >> 2)][m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bo.v[m & 3], ao.v[m >> 2], acc[4 + (m >> 2)][m & 3], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      pb = sa;
      pa = sb;
      sa = sa1;
      sb = sb1;
    };

    const int np = p.stages;
    for (int kc = 0; kc < mine; ++kc) {
      // quarter 0's fragments of the tile's first stage
      read_frag(I0{}, I0{}, sa, sb, ae, be);
      read_frag(I0{}, I1{}, sa, sb, ae, be);
      read_frag(I0{}, I2{}, sa, sb, ae, be);
      read_frag(I0{}, I3{}, sa, sb, ae, be);
      read_frag(I0{}, std::integral_constant<int, 4>{}, sa, sb, ae, be);
      read_frag(I0{}, std::integral_constant<int, 5>{}, sa, sb, ae, be);
      read_frag(I0{}, std::integral_constant<int, 6>{}, sa, sb, ae, be);
      read_frag(I0{}, std::integral_constant<int, 7>{}, sa, sb, ae, be);
      retire(ae);
      retire(be);
      TN_PIN();
      for (int t = 1; t < np; ++t) trip(F_{});
      trip(T_{});
      int m0, n0;
      origin(kc, m0, n0);
      char* park = smem + (wave < 4 ? pb : pa) * SLOT + (wave & 3) * 8192;
      if constexpr (EPI == EPI_SWIGLU_FWD) {
        epilogue16_swiglu_fwd(p, acc, park, m0 + wr * 128, n0 + wc * 32, lane);
      } else if constexpr (EPI == EPI_SWIGLU_BWD) {
        epilogue16_swiglu_bwd(p, acc, park, m0 + wr * 128, n0 + wc * 64, lane);
      } else if constexpr (EPI == EPI_ROPE) {
        const bool d128 = p.rope_d == 128;
        epilogue16_rope(p, acc, park, m0 + wr * 128, d128 ? n0 + (wc >> 1) * 128 + (wc & 1) * 32 : n0 + wc * 64,
                        d128 ? (wc & 1) * 32 : 0, lane);
      } else {
        epilogue16(p, acc, park, m0 + wr * 128, n0 + wc * 64, lane);
      }
      zero_acc();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      asm volatile("" : "+v"(lane));
      ra = Reader<AK, 8>(lane, wr * 128);
      rb = Reader<BK, 4>(lane, wc * 64);
      sA.set_voff(wave, lane);
      sB.set_voff(b_wave0 ? 0 : wave, lane);
      early_pieces(pb, pa);
    }
  }

  // ---- park helpers: a 64-row half of the wave tile (A blocks 4 half .. 4 half + 3) <-> the [64 rows][64 columns] bf16 park.
  // Element (row = 16 bi + l15, columns 16 bj + 4 g4 .. + 3) = one 8-byte pack at chunk (2 bj + (g4 >> 1)) ^ (row & 7).
  static __device__ __forceinline__ char* park_at(char* park, int bi, int bj, int l15, int g4) {
    const int row = bi * 16 + l15;
    return park + row * 128 + (((bj * 2 + (g4 >> 1)) ^ (row & 7)) << 4) + (g4 & 1) * 8;
  }

  // SwiGLU forward (see Kernel::epilogue_swiglu_fwd): blocks bj = 0, 1 of a lane are gate, bj = 2, 3 up of the SAME columns
  // wn0 + 16 bj + 4 g4 ..; three park trips: gate | up of 64 rows each, then act of all 128 rows.
  static __device__ __forceinline__ void epilogue16_swiglu_fwd(const Params& p, Acc16& acc, char* park, int wm0, int wn0,
                                                               int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
    const uintptr_t c_gate = (uintptr_t)p.C, c_up = (uintptr_t)p.C2;
    bf16_t* const gu_base = reinterpret_cast<bf16_t*>(c_gate + ((lane & 4) ? c_up - c_gate : (uintptr_t)0));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
          const f32x4_t a = acc[half * 4 + bi][bj];
          *reinterpret_cast<u32x2_t*>(park_at(park, bi, bj, l15, g4)) = u32x2_t{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
        const int m = wm0 + half * 64 + row, n = wn0 + (c & 3) * 8;
        if (m < p.M && n < p.N)
          *reinterpret_cast<uint4*>(gu_base + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
      }
    }
#pragma unroll
    for (int bi = 0; bi < 8; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj) {
        float h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gf = rbf(acc[bi][bj][e]), uf = rbf(acc[bi][bj + 2][e]);
          h[e] = rbf(gf * sigm(gf)) * uf;
        }
        // columns 0-31 of the park = rows 0-63, columns 32-63 = rows 64-127
        *reinterpret_cast<u32x2_t*>(park_at(park, bi & 3, (bi >> 2) * 2 + bj, l15, g4)) =
            u32x2_t{pack2bf(h[0], h[1]), pack2bf(h[2], h[3])};
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), c = lane & 7;
      const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
      const int m = wm0 + (c >> 2) * 64 + row, n = wn0 + (c & 3) * 8;
      if (m < p.M && n < p.N)
        *reinterpret_cast<uint4*>(p.C3 + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
    }
  }

  // SwiGLU backward (see Kernel::epilogue_swiglu_bwd): gate / up rows in through the park, d(gate) / d(up) out through it
  static __device__ __forceinline__ void epilogue16_swiglu_bwd(const Params& p, Acc16& acc, char* park, int wm0, int wn0,
                                                               int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 gv[8], uv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const int m = wm0 + half * 64 + row, n = wn0 + c * 8;
        const bool in = m < p.M && n < p.N;
        const long long off = (long long)m * p.lde + n;
        gv[it] = in ? *reinterpret_cast<const uint4*>(p.E1 + off) : make_uint4(0, 0, 0, 0);
        uv[it] = in ? *reinterpret_cast<const uint4*>(p.E2 + off) : make_uint4(0, 0, 0, 0);
      }
      u32x2_t gq[4][4], uq[4][4];
      auto through_park = [&](const uint4 (&src)[8], u32x2_t (&dst)[4][4]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), c = lane & 7;
          const u32x4_t v = {src[it].x, src[it].y, src[it].z, src[it].w};
          *reinterpret_cast<u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4)) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) dst[bi][bj] = *reinterpret_cast<const u32x2_t*>(park_at(park, bi, bj, l15, g4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      through_park(gv, gq);
      through_park(uv, uq);
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
          float dg[4], du[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t wg = e < 2 ? gq[bi][bj].x : gq[bi][bj].y, wu = e < 2 ? uq[bi][bj].x : uq[bi][bj].y;
            const float gf = (e & 1) ? hi16(wg) : lo16(wg), uf = (e & 1) ? hi16(wu) : lo16(wu);
            const float d = rbf(acc[half * 4 + bi][bj][e]);
            const float sg = sigm(gf);
            const float silu = gf * sg;
            du[e] = d * silu;
            dg[e] = d * uf * (sg + silu * (1.f - sg));
          }
          gq[bi][bj] = u32x2_t{pack2bf(dg[0], dg[1]), pack2bf(dg[2], dg[3])};
          uq[bi][bj] = u32x2_t{pack2bf(du[0], du[1]), pack2bf(du[2], du[3])};
        }
      auto out_park = [&](const u32x2_t (&src)[4][4], bf16_t* C) {
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
          for (int bj = 0; bj < 4; ++bj) *reinterpret_cast<u32x2_t*>(park_at(park, bi, bj, l15, g4)) = src[bi][bj];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
        for (int it = 0; it < 8; ++it) {
          const int row = it * 8 + (lane >> 3), c = lane & 7;
          const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
          const int m = wm0 + half * 64 + row, n = wn0 + c * 8;
          if (m < p.M && n < p.N)
            *reinterpret_cast<uint4*>(C + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
        }
      };
      out_park(gq, p.C);
      out_park(uq, p.C2);
    }
  }

  // RoPE (see Kernel::epilogue_rope): blocks bj = 0, 1 hold columns col_a + 16 bj + 4 g4 .., blocks bj = 2, 3 the partners
  // half a head further (col_a + D / 2 + ..); c0 = index of col_a's pair inside the head
  static __device__ __forceinline__ void epilogue16_rope(const Params& p, Acc16& acc, char* park, int wm0, int col_a, int c0,
                                                         int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
    const int hd = p.rope_d >> 1;
    const int col_b = col_a + hd;
    uint2 bwa[2], bwb[2];
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
      const int q = 16 * bj + 4 * g4;
      bwa[bj] = bwb[bj] = make_uint2(0, 0);
      if (p.bias != nullptr) {
        bwa[bj] = *reinterpret_cast<const uint2*>(p.bias + min(col_a + q, p.N - 4));
        bwb[bj] = *reinterpret_cast<const uint2*>(p.bias + min(col_b + q, p.N - 4));
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint2 cw[4][2], sw[4][2];
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        const int m = min(wm0 + (half * 4 + bi) * 16 + l15, p.M - 1);
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          const long long o = (long long)m * hd + c0 + 16 * bj + 4 * g4;
          cw[bi][bj] = *reinterpret_cast<const uint2*>(p.rope_cos + o);
          sw[bi][bj] = *reinterpret_cast<const uint2*>(p.rope_sin + o);
        }
      }
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
          const float ba[4] = {lo16(bwa[bj].x), hi16(bwa[bj].x), lo16(bwa[bj].y), hi16(bwa[bj].y)};
          const float bb[4] = {lo16(bwb[bj].x), hi16(bwb[bj].x), lo16(bwb[bj].y), hi16(bwb[bj].y)};
          const float cf[4] = {lo16(cw[bi][bj].x), hi16(cw[bi][bj].x), lo16(cw[bi][bj].y), hi16(cw[bi][bj].y)};
          const float sf[4] = {lo16(sw[bi][bj].x), hi16(sw[bi][bj].x), lo16(sw[bi][bj].y), hi16(sw[bi][bj].y)};
          float ya[4], yb[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            rope_rotate(rbf(acc[half * 4 + bi][bj][e] + ba[e]), rbf(acc[half * 4 + bi][bj + 2][e] + bb[e]), cf[e], sf[e],
                        ya[e], yb[e]);
          *reinterpret_cast<u32x2_t*>(park_at(park, bi, bj, l15, g4)) = u32x2_t{pack2bf(ya[0], ya[1]), pack2bf(ya[2], ya[3])};
          *reinterpret_cast<u32x2_t*>(park_at(park, bi, bj + 2, l15, g4)) = u32x2_t{pack2bf(yb[0], yb[1]), pack2bf(yb[2], yb[3])};
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
        const int m = wm0 + half * 64 + row, n = (c < 4 ? col_a : col_b) + (c & 3) * 8;
        if (m < p.M && n < p.N)
          *reinterpret_cast<uint4*>(p.C + (long long)m * p.ldc + n) = make_uint4(pv.x, pv.y, pv.z, pv.w);
      }
    }
  }

  // Epilogue through LDS, as Kernel::epilogue: the wave parks its 128 x 64 tile 64 rows at a time in its XOR-swizzled 8 KB
  // ([64 rows m][64 columns n] bf16, 16-byte chunk ^= row & 7) and writes full 128-byte lines.  A lane holds, per 16 x 16
  // block, column m = l & 15 and the 4 consecutive n = 4 (l >> 4) ..: 8-byte packs.
  static __device__ __forceinline__ void epilogue16(const Params& p, Acc16& acc, char* park, int wm0, int wn0, int lane) {
    const int l15 = lane & 15, g4 = lane >> 4;
    float bias_v[4][4];
    if (p.bias != nullptr) {
#pragma unroll
      for (int bj = 0; bj < 4; ++bj) {
        const int n = min(wn0 + bj * 16 + 4 * g4, p.N - 4);
        const uint2 w = *reinterpret_cast<const uint2*>(p.bias + n);
        bias_v[bj][0] = __uint_as_float(w.x << 16);
        bias_v[bj][1] = __uint_as_float(w.x & 0xffff0000u);
        bias_v[bj][2] = __uint_as_float(w.y << 16);
        bias_v[bj][3] = __uint_as_float(w.y & 0xffff0000u);
      }
    }
    const bool acc_c = p.accumulate != 0 || p.addend != nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // GELU backward: the pre-activation rows this lane will meet in the read-out loop, asked for before the tile is parked
      // (fetched row by row inside that loop every load's latency was exposed: one workgroup per CU, attn_bwd_dq_stream.hip)
      uint4 xpre[8];
      if constexpr (EPI == EPI_GELU_BWD) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int m = wm0 + half * 64 + it * 8 + (lane >> 3), n = wn0 + (lane & 7) * 8;
          xpre[it] = (m < p.M && n < p.N) ? *reinterpret_cast<const uint4*>(p.E1 + (long long)m * p.lde + n)
                                          : make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        const int row = bi * 16 + l15;
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[half * 4 + bi][bj][e] + (p.bias != nullptr ? bias_v[bj][e] : 0.f);
          const int chunk = (bj * 2 + (g4 >> 1)) ^ (row & 7);
          const u32x2_t pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2_t*>(park + row * 128 + chunk * 16 + (g4 & 1) * 8) = pk;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int kReadOutUnroll = EPI == EPI_GELU_BWD ? 8 : 4;       // (xpre[it] must be a register, not an indexed array)
#pragma unroll kReadOutUnroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), c = lane & 7;
        const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(park + row * 128 + ((c ^ (row & 7)) << 4));
        uint4 v = make_uint4(pv.x, pv.y, pv.z, pv.w);
        const int m = wm0 + half * 64 + row, n = wn0 + c * 8;
        if (m < p.M && n < p.N) {
          bf16_t* dst = p.C + (long long)m * p.ldc + n;
          if constexpr (EPI == EPI_GELU_FWD) {
            // v = the rounded pre-activation: stored, and its GELU beside it (what the activation kernel made of it)
            *reinterpret_cast<uint4*>(dst) = v;
            Vec16<bf16_t> x;
            x.raw = v;
            float f[8];
            x.unpack(f);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const gelu_f32x2 r = gelu_f2(gelu_f32x2{f[e], f[e + 1]});
              f[e] = r.x;
              f[e + 1] = r.y;
            }
            x.pack(f);
            *reinterpret_cast<uint4*>(p.C2 + (long long)m * p.ldc + n) = x.raw;
            continue;
          }
          if constexpr (EPI == EPI_GELU_BWD) {
            // v = the rounded d(act): times gelu'(pre), what the GELU backward kernel made of the two
            Vec16<bf16_t> x, d;
            x.raw = xpre[it];
            d.raw = v;
            float f[8], df[8];
            x.unpack(f);
            d.unpack(df);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const gelu_f32x2 r = gelu_grad_f2(gelu_f32x2{f[e], f[e + 1]}, gelu_f32x2{df[e], df[e + 1]});
              df[e] = r.x;
              df[e + 1] = r.y;
            }
            d.pack(df);
            *reinterpret_cast<uint4*>(dst) = d.raw;
            continue;
          }
          if (acc_c) {
            Vec16<bf16_t> o, nw;
            o.load(p.addend != nullptr ? p.addend + (long long)m * p.ldadd + n : dst);
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
    }
  }
};

template <bool AK, bool BK, int PLACE, int EPI = EPI_PLAIN>
__global__ __launch_bounds__(NT, 2) void gemm16_kernel(const Params p) {
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
  Kernel16<AK, BK, PLACE, EPI>::run(p, smem);
}

// ws[S][ntiles][256 x 256] fp32 partial sums -> C = bf16(sum_s ws[s] (+ bias) (+ C)) on the tiles [tile0, tile0 + ntiles);
// 32 blocks of 256 threads per tile, one thread per 8 consecutive columns
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int M, int N, int nbm,
                                                            int nbn, int tile0, int ntiles, bf16_t* __restrict__ C,
                                                            long long ldc, const bf16_t* __restrict__ bias,
                                                            int accumulate, int out_f32,
                                                            const float* __restrict__ bias_ws,
                                                            bf16_t* __restrict__ bias_out) {
  const int t = blockIdx.x >> 5, part = blockIdx.x & 31;
  int tm, tn;
  tile_of_block(tile0 + t, nbm, nbn, tm, tn);
  const int idx = part * 256 + threadIdx.x;                 // 0 .. 8191: (row, 8-column group) of the tile
  const int lm = idx >> 5, ln = (idx & 31) * 8;
  const int m = tm * BM + lm, n = tn * BN + ln;
  if (m >= M || n >= N) return;
  if (bias_out != nullptr && tn == 0 && ln == 0) {          // EPI_BIASG: the units' partial column sums of A, row m
    float b = 0.f;
    for (int sp = 0; sp < S; ++sp) b += bias_ws[(long long)sp * (nbm * BM) + m];
    bias_out[m] = f2bf(b);
  }
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* src = ws + (long long)t * (BM * BN) + lm * BN + ln;
  const long long slab = (long long)ntiles * (BM * BN);
  for (int sp = 0; sp < S; ++sp) {
    const float4 x = *reinterpret_cast<const float4*>(src + sp * slab);
    const float4 y = *reinterpret_cast<const float4*>(src + sp * slab + 4);
    a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
    a[4] += y.x; a[5] += y.y; a[6] += y.z; a[7] += y.w;
  }
  if (out_f32) {                                           // (weight gradients into fp32 staging: no bias)
    float* d = reinterpret_cast<float*>(C) + (long long)m * ldc + n;
    float4 x = make_float4(a[0], a[1], a[2], a[3]), y = make_float4(a[4], a[5], a[6], a[7]);
    if (accumulate) {
      const float4 ox = *reinterpret_cast<const float4*>(d), oy = *reinterpret_cast<const float4*>(d + 4);
      x.x += ox.x; x.y += ox.y; x.z += ox.z; x.w += ox.w;
      y.x += oy.x; y.y += oy.y; y.z += oy.z; y.w += oy.w;
    }
    *reinterpret_cast<float4*>(d) = x;
    *reinterpret_cast<float4*>(d + 4) = y;
    return;
  }
  bf16_t* dst = C + (long long)m * ldc + n;
  if (bias != nullptr) {
    Vec16<bf16_t> b;
    float fb[8];
    b.load(bias + n);
    b.unpack(fb);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += fb[e];
  }
  if (accumulate) {
    Vec16<bf16_t> o;
    float fo[8];
    o.load(dst);
    o.unpack(fo);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += fo[e];
  }
  Vec16<bf16_t> out;
  out.pack(a);
  out.store(dst);
}

// kernel-development switch: TN_GEMM_VARIANT = 100 * PLACE + 10 * ASYM + ILV (scripts/gemm_sweep.py sweeps it)
#ifndef TN_GEMM_DEFAULT_VARIANT
#define TN_GEMM_DEFAULT_VARIANT 601
#endif

// The half-stage ring (Kernel32) and the four-wave / AGPR-accumulator geometry (Kernel4) are measured variants, not product:
// scripts/variants/gemm_kernel32_kernel4.inc, compiled in only by variant builds.
#if defined(TN_GEMM_ALL_VARIANTS) || TN_GEMM_DEFAULT_VARIANT >= 1000
#define TN_GEMM_HAVE_VARIANT_KERNELS 1
#include "../../scripts/variants/gemm_kernel32_kernel4.inc"
#endif


template <bool AK, bool BK, bool HAS_CT>
static int launch_variant(int variant, dim3 grid, hipStream_t st, Params p) {
  const int stages64 = p.stages;
  p.stages = 2 * stages64;                     // (the half-stage ring counts 32-deep stages; reset below for the others)
#define TN_V(PL, AS, IL)                                                                             \
  case 100 * PL + 10 * AS + IL:                                                                      \
    p.stages = stages64;                                                                             \
    hipLaunchKernelGGL((gemm_kernel<AK, BK, PL, AS, IL, HAS_CT>), grid, dim3(NT), 0, st, p);         \
    return 0;
  constexpr int DPL = (TN_GEMM_DEFAULT_VARIANT % 1000) / 100, DAS = (TN_GEMM_DEFAULT_VARIANT / 10) % 10,
                DIL = TN_GEMM_DEFAULT_VARIANT % 10;
#define TN_V32(PL)                                                                                   \
  case 1000 + PL:                                                                                    \
    hipLaunchKernelGGL((gemm32_kernel<AK, BK, PL, HAS_CT>), grid, dim3(NT), 0, st, p);               \
    return 0;
#define TN_V4(DN)                                                                                    \
  case 2000 + DN:                                                                                    \
    p.stages = stages64;                                                                             \
    hipLaunchKernelGGL((gemm4_kernel<AK, BK, DN, HAS_CT>), grid, dim3(256), 0, st, p);               \
    return 0;
#ifdef TN_GEMM_ALL_VARIANTS
  if constexpr (!HAS_CT) {
    switch (variant) {
      TN_V4(1) TN_V4(2) TN_V4(3)
      TN_V32(0) TN_V32(1) TN_V32(2) TN_V32(3)
      TN_V(0, 0, 0) TN_V(0, 0, 1)
      TN_V(1, 0, 0) TN_V(1, 0, 1) TN_V(1, 1, 0) TN_V(1, 1, 1)
      TN_V(3, 0, 1) TN_V(3, 1, 1)
      TN_V(4, 0, 1) TN_V(4, 1, 1)
      TN_V(5, 0, 1) TN_V(5, 1, 1)
      TN_V(6, 0, 1) TN_V(6, 1, 1)
      default:
        return -1;
    }
  }
#endif
  (void)variant;
#if TN_GEMM_DEFAULT_VARIANT >= 2000
  p.stages = stages64;
  hipLaunchKernelGGL((gemm4_kernel<AK, BK, TN_GEMM_DEFAULT_VARIANT - 2000, HAS_CT>), grid, dim3(256), 0, st, p);
#elif TN_GEMM_DEFAULT_VARIANT >= 1000
  hipLaunchKernelGGL((gemm32_kernel<AK, BK, TN_GEMM_DEFAULT_VARIANT - 1000, HAS_CT>), grid, dim3(NT), 0, st, p);
#else
  {
    p.stages = stages64;
    if (p.bias_out != nullptr) {                   // weight gradient + bias gradient (EPI_BIASG)
      if constexpr (AK && BK && !HAS_CT) {
        if (p.splitk > 1) {
          hipLaunchKernelGGL((gemm_kernel<true, true, DPL, DAS, DIL, false, true, false, EPI_BIASG>), grid, dim3(NT), 0, st, p);
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)p.ntiles * 32u), dim3(256), 0, st, p.ws, p.splitk, p.M,
                             p.N, p.nbm, p.nbn, p.tile0, p.ntiles, p.C, p.ldc, p.bias, p.accumulate, p.c_f32,
                             (const float*)p.bias_ws, p.bias_out);
        } else if (p.c_f32) {
          hipLaunchKernelGGL((gemm_kernel<true, true, DPL, DAS, DIL, false, false, true, EPI_BIASG>), grid, dim3(NT), 0, st, p);
        } else {
          hipLaunchKernelGGL((gemm_kernel<true, true, DPL, DAS, DIL, false, false, false, EPI_BIASG>), grid, dim3(NT), 0, st, p);
        }
      } else {
        return -1;
      }
    } else if (p.splitk > 1) {
      if constexpr (!HAS_CT) {
        hipLaunchKernelGGL((gemm_kernel<AK, BK, DPL, DAS, DIL, false, true>), grid, dim3(NT), 0, st, p);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)p.ntiles * 32u), dim3(256), 0, st, p.ws, p.splitk, p.M,
                           p.N, p.nbm, p.nbn, p.tile0, p.ntiles, p.C, p.ldc, p.bias, p.accumulate, p.c_f32,
                           (const float*)nullptr, (bf16_t*)nullptr);
      } else {
        return -1;
      }
    } else if (p.c_f32) {
      if constexpr (AK && BK && !HAS_CT) {         // (weight-gradient mode only: the one caller)
        hipLaunchKernelGGL((gemm_kernel<true, true, DPL, DAS, DIL, false, false, true>), grid, dim3(NT), 0, st, p);
      } else {
        return -1;
      }
    } else {
      // row-stored A (forward / input-gradient products): the 16x16x32 kernel (TN_GEMM_M16=0: the 32x32x16 one)
      static const bool m16 = [] { const char* e = getenv("TN_GEMM_M16"); return !(e && e[0] == '0'); }();
      if constexpr (!HAS_CT && !AK) {
        if (m16) {
#ifdef TN_G16_SWEEP   // kernel development: DMA placement table of Kernel16 from the environment (scripts/r05_g16_sweep.sh)
          const char* e16 = getenv("TN_G16_PLACE");
          switch (e16 ? atoi(e16) : DPL) {
            case 1: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 1>), grid, dim3(NT), 0, st, p); return 0;
            case 2: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 2>), grid, dim3(NT), 0, st, p); return 0;
            case 3: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 3>), grid, dim3(NT), 0, st, p); return 0;
            case 4: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 4>), grid, dim3(NT), 0, st, p); return 0;
            case 5: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 5>), grid, dim3(NT), 0, st, p); return 0;
            case 0: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 0>), grid, dim3(NT), 0, st, p); return 0;
            case 7: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 7>), grid, dim3(NT), 0, st, p); return 0;
            case 8: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 8>), grid, dim3(NT), 0, st, p); return 0;
            case 9: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 9>), grid, dim3(NT), 0, st, p); return 0;
            case 10: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 10>), grid, dim3(NT), 0, st, p); return 0;
            case 11: hipLaunchKernelGGL((gemm16_kernel<AK, BK, 11>), grid, dim3(NT), 0, st, p); return 0;
            default: break;
          }
#endif
          hipLaunchKernelGGL((gemm16_kernel<AK, BK, DPL>), grid, dim3(NT), 0, st, p);
          return 0;
        }
      }
      hipLaunchKernelGGL((gemm_kernel<AK, BK, DPL, DAS, DIL, HAS_CT>), grid, dim3(NT), 0, st, p);
    }
  }
#endif
  return 0;
#undef TN_V
#undef TN_V32
#undef TN_V4
}

}  // namespace gemm
}  // namespace tn

extern "C" {

// One workgroup per CU walking a fixed tile list (1, default) or one workgroup per tile (0).  A host that runs collectives
// beside the compute (RCCL kernels hold CUs) selects 0 — touchnet_amd/bin/train.py — instead of changing the process
// environment; TN_GEMM_PERSIST in the environment still overrides it.
static int g_persistent = 1;
void tn_gemm_set_persistent(int on) { g_persistent = on ? 1 : 0; }
int tn_gemm_get_persistent(void) { return g_persistent; }

// General entry: C[M,N] = sum_s opA_s · opB_s^T (+ bias) (+ C if accumulate); optional transposed copy Ct[N,M].
//   a_kmaj / b_kmaj: 0 = operand stored [rows, K] (contraction-contiguous), 1 = stored [K, rows] (contraction-major).
//   A, B, lda, ldb, K: arrays of nseg (1..3) entries.
// Requirements (else -22): every K % 64 == 0 (any K when both operands are contraction-major), N % 8 == 0, ld % 8 == 0, 16-byte aligned bases; KMAJ operands span
// < 2 GB ((K-1) * ld + rows) * 2 bytes); with Ct: M % 8 == 0, ldct % 8 == 0.
static int gemm_launch(const void* const* A, const void* const* B, const long long* lda, const long long* ldb,
                       const int* K, int nseg, int a_kmaj, int b_kmaj, void* C, void* Ct, const void* bias, int M, int N,
                       long long ldc, long long ldct, int accumulate, int splitk, int tail_only, void* workspace,
                       long long workspace_bytes, void* stream, int c_f32 = 0, void* bias_grad = nullptr,
                       const void* addend = nullptr, long long ldadd = 0) {
  using namespace tn::gemm;
  // (the addend rides on the plain epilogue of an unsplit, single-output product)
  if (addend != nullptr && (accumulate || splitk > 1 || tail_only || Ct != nullptr || c_f32 || bias_grad != nullptr ||
                            (ldadd % 8) || ldadd < N || ((uintptr_t)addend & 15)))
    return TN_EINVAL;
  if (bias_grad != nullptr && (!(a_kmaj && b_kmaj) || nseg != 1 || Ct != nullptr || bias != nullptr || tail_only ||
                               ((uintptr_t)bias_grad & 1) || TN_GEMM_DEFAULT_VARIANT >= 1000 ||
                               getenv("TN_GEMM_VARIANT") != nullptr))
    return TN_EINVAL;
  if (M <= 0 || N <= 0 || nseg < 1 || nseg > MAXSEG || (N % 8) != 0) return TN_EINVAL;
  if ((ldc % 8) || ldc < N || ((uintptr_t)C & 15)) return TN_EINVAL;
  if (c_f32 && (!(a_kmaj && b_kmaj) || nseg != 1 || Ct != nullptr || bias != nullptr || tail_only ||
                TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr))
    return TN_EINVAL;
  if (a_kmaj && (M % 8)) return TN_EINVAL;
  Params p;
  p = Params{};
  p.stages = 0;
  for (int s = 0; s < nseg; ++s) {
    const int k = K[s];
    // (two contraction-major operands: rows k >= K are zero-filled by both descriptors, so any depth works; a
    //  contraction-contiguous operand would read its own next columns there)
    if (k <= 0 || ((k % 64) != 0 && !(a_kmaj && b_kmaj)) || (lda[s] % 8) || (ldb[s] % 8)) return TN_EINVAL;
    if (((uintptr_t)A[s] | (uintptr_t)B[s]) & 15) return TN_EINVAL;
    if (a_kmaj) {
      if (lda[s] < M || ((long long)(k - 1) * lda[s] + M) * 2 >= 0x7fffffffLL) return TN_EINVAL;
    } else {
      // per-tile DMA offsets are 32-bit: 288 rows of the operand must stay below 2 GB.  (lda < K is allowed: OVERLAPPING rows,
      // the im2col view of a k = 3 convolution over a channels-last sequence — row r = 3 C elements from r * stride * C on)
      if (lda[s] <= 0 || (long long)288 * lda[s] * 2 >= 0x7fffffffLL) return TN_EINVAL;
    }
    if (b_kmaj) {
      if (ldb[s] <= 0 || ((long long)(k - 1) * ldb[s] + N) * 2 >= 0x7fffffffLL) return TN_EINVAL;   // (ldb < N: overlapping rows)
    } else {
      if (ldb[s] <= 0 || (long long)288 * ldb[s] * 2 >= 0x7fffffffLL) return TN_EINVAL;
    }
    p.seg[s].A = (const tn::bf16_t*)A[s];
    p.seg[s].B = (const tn::bf16_t*)B[s];
    p.seg[s].lda = lda[s];
    p.seg[s].ldb = ldb[s];
    p.seg[s].K = k;
    p.seg[s].pad_ = 0;
    p.stages += (k + 63) / 64;
  }
  for (int s = nseg; s < MAXSEG; ++s) p.seg[s] = p.seg[0];
  if (Ct != nullptr && ((M % 8) || (ldct % 8) || ldct < M || ((uintptr_t)Ct & 15))) return TN_EINVAL;
  if (Ct != nullptr && accumulate) return TN_EINVAL;
  p.nseg = nseg;
  p.M = M;
  p.N = N;
  p.C = (tn::bf16_t*)C;
  p.Ct = (tn::bf16_t*)Ct;
  p.bias = (const tn::bf16_t*)bias;
  p.ldc = ldc;
  p.ldct = ldct;
  p.accumulate = accumulate;
  p.addend = (const tn::bf16_t*)addend;
  p.ldadd = ldadd;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (N + BN - 1) / BN;
  p.splitk = 1;
  p.kchunk = 0;
  p.ws = nullptr;
  p.tile0 = 0;
  p.ntiles = p.nbm * p.nbn;
  p.c_f32 = c_f32;
  // persistent: one workgroup per CU walks its tiles (TN_GEMM_PERSIST=0: one workgroup per tile, kernel-development A/B)
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? n / 8 * 8 : 8;
  }();
  int main_tiles = 0;                                // tail split: tiles [0, main_tiles) run unsplit first
  if (splitk > 1) {
    // one segment, no transposed copy; a contraction-contiguous operand cannot be cut inside a stage or read behind K
    if (nseg != 1 || Ct != nullptr || TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr)
      return TN_EINVAL;
    if (!(a_kmaj && b_kmaj) && (p.stages % splitk) != 0) return TN_EINVAL;
    if (tail_only) {
      main_tiles = p.ntiles / ncu * ncu;
      if (main_tiles <= 0 || main_tiles >= p.ntiles) return TN_EINVAL;
    }
    const long long slabs = (long long)splitk * (p.ntiles - main_tiles) * BM * BN;      // floats
    if (workspace == nullptr || ((uintptr_t)workspace & 15) ||
        workspace_bytes < (slabs + (bias_grad ? (long long)splitk * p.nbm * BM : 0)) * (long long)sizeof(float))
      return TN_EINVAL;
    p.splitk = splitk;
    p.kchunk = (p.stages + splitk - 1) / splitk;
    p.ws = (float*)workspace;
    p.bias_ws = bias_grad ? (float*)workspace + slabs : nullptr;     // [splitk][nbm * 256] partial column sums
  }
  p.bias_out = (tn::bf16_t*)bias_grad;
  const char* pe = getenv("TN_GEMM_PERSIST");        // (the environment wins: kernel-development A/B)
  const bool persist = pe ? atoi(pe) != 0 : g_persistent != 0;
  hipStream_t st = (hipStream_t)stream;
  const char* e = getenv("TN_GEMM_VARIANT");     // kernel-development A/B switch (read per call: the sweep changes it)
  const int variant = e ? atoi(e) : TN_GEMM_DEFAULT_VARIANT;
  int rc = 0;
#define TN_MODE(AKM, BKM, GRID, PRM)                                                       \
  rc |= (Ct != nullptr) ? launch_variant<AKM, BKM, true>(variant, GRID, st, PRM)           \
                        : launch_variant<AKM, BKM, false>(variant, GRID, st, PRM)
#define TN_LAUNCH(GRID, PRM)                                                               \
  if (!a_kmaj && !b_kmaj) TN_MODE(false, false, GRID, PRM);                                \
  else if (!a_kmaj && b_kmaj) TN_MODE(false, true, GRID, PRM);                             \
  else if (a_kmaj && b_kmaj) TN_MODE(true, true, GRID, PRM);                               \
  else return TN_EINVAL /* (A contraction-major with B contraction-contiguous: no caller) */
  if (main_tiles > 0) {                          // whole rounds first, unsplit ...
    Params pm = p;
    pm.splitk = 1;
    pm.ntiles = main_tiles;
    const dim3 gm(ncu);
    TN_LAUNCH(gm, pm);
    p.tile0 = main_tiles;                        // ... then the last partial round, split
    p.ntiles -= main_tiles;
  }
  {
    const int units = p.ntiles * p.splitk;
    const dim3 grid(persist && units > ncu ? ncu : units);
    TN_LAUNCH(grid, p);
  }
#undef TN_LAUNCH
#undef TN_MODE
  if (rc != 0) return TN_EINVAL;
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_gemm_bf16(const void* const* A, const void* const* B, const long long* lda, const long long* ldb, const int* K,
                 int nseg, int a_kmaj, int b_kmaj, void* C, void* Ct, const void* bias, int M, int N, long long ldc,
                 long long ldct, int accumulate, void* stream) {
  return gemm_launch(A, B, lda, ldb, K, nseg, a_kmaj, b_kmaj, C, Ct, bias, M, N, ldc, ldct, accumulate, 1, 0, nullptr, 0,
                     stream);
}

// C = A B^T (+ bias) + addend: the residual stream added in the producing GEMM's epilogue (addend [M, N] bf16, pitch ldadd;
// it may NOT alias C).  One segment; the product is rounded to bf16 before the addition — the bits of the two-kernel form
// (GEMM, then the norm kernel's residual add: modeling_llama.py / modeling_qwen2.py `hidden_states = residual +
// hidden_states`).
int tn_gemm_bf16_addend(const void* A, const void* B, long long lda, long long ldb, int K, int a_kmaj, int b_kmaj, void* C,
                        const void* bias, const void* addend, long long ldadd, int M, int N, long long ldc, void* stream) {
  if (addend == nullptr || addend == C) return TN_EINVAL;
  return gemm_launch(&A, &B, &lda, &ldb, &K, 1, a_kmaj, b_kmaj, C, nullptr, bias, M, N, ldc, 0, 0, 1, 0, nullptr, 0, stream,
                     0, nullptr, addend, ldadd);
}

// The same product with the contraction cut into `splitk` parts that run as independent units; fp32 partial sums go through
// `workspace` (whole 256 x 256 tile slabs: >= splitk * tiles_split * 262144 bytes, 16-byte aligned) and a second kernel
// adds them up (+ bias, + C).  tail_only = 0: every tile is split (outputs of few tiles with a deep contraction);
// tail_only = 1: the whole rounds of tiles (floor(tiles / CUs) * CUs) run unsplit, only the last partial round is split
// (tiles_split = tiles mod CUs; -22 when there is no whole round or no remainder).  One segment, no transposed copy; with a
// contraction-contiguous operand the number of 64-deep stages must be a multiple of splitk (else -22).
int tn_gemm_bf16_splitk(const void* A, const void* B, long long lda, long long ldb, int K, int a_kmaj, int b_kmaj,
                        void* C, const void* bias, int M, int N, long long ldc, int accumulate, int splitk, int tail_only,
                        void* workspace, long long workspace_bytes, void* stream) {
  if (splitk < 2) return TN_EINVAL;
  return gemm_launch(&A, &B, &lda, &ldb, &K, 1, a_kmaj, b_kmaj, C, nullptr, bias, M, N, ldc, 0, accumulate, splitk,
                     tail_only, workspace, workspace_bytes, stream);
}

// Weight gradient with fp32 output: C[M,N] (float, ldc in floats) = (+=) A[K,M]^T · B[K,N], both operands contraction-major
// (dW = dY^T x read as stored).  splitk <= 1: plain; >= 2: split-K through `workspace` as in tn_gemm_bf16_splitk.
int tn_gemm_bf16_wgrad_f32(const void* A, const void* B, long long lda, long long ldb, int K, float* C, int M, int N,
                           long long ldc, int accumulate, int splitk, void* workspace, long long workspace_bytes,
                           void* stream) {
  return gemm_launch(&A, &B, &lda, &ldb, &K, 1, 1, 1, C, nullptr, nullptr, M, N, ldc, 0, accumulate,
                     splitk > 1 ? splitk : 1, 0, workspace, workspace_bytes, stream, 1);
}

namespace {

// 16x16x32 MFMAs (Kernel16) for the products whose A operand is row-stored — forward and input-gradient products — where
// they are 5-7 % / 2-3 % faster; the weight-gradient mode (both operands through transpose reads) gains nothing and stays on
// Kernel.  TN_GEMM_M16=0: every product on the 32x32x16 kernel (A/B runs).
bool use_m16() {
  static const bool on = [] { const char* e = getenv("TN_GEMM_M16"); return !(e && e[0] == '0'); }();
  return on;
}

constexpr int kDPL = (TN_GEMM_DEFAULT_VARIANT % 1000) / 100, kDAS = (TN_GEMM_DEFAULT_VARIANT / 10) % 10,
              kDIL = TN_GEMM_DEFAULT_VARIANT % 10;

int num_cus() {
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n >= 8 ? n / 8 * 8 : 8;
  }();
  return ncu;
}

bool persistent_now() {
  const char* pe = getenv("TN_GEMM_PERSIST");
  return pe ? atoi(pe) != 0 : g_persistent != 0;
}

void clear_params(tn::gemm::Params& p) {
  p = tn::gemm::Params{};
  p.splitk = 1;
  p.nseg = 1;
}

}  // namespace

// Several independent products of ONE operand mode in one persistent launch (`ngrp` = 1..3; product g:
// C_g[M_g, N_g] (+)= opA_g · opB_g^T, one segment each, no bias).  What it is for: the MLP's three weight gradients are 688
// output tiles each = 2.69 rounds on 256 CUs, i.e. three launches idle a third of the chip in their last rounds; as one
// tile list they are 2064 tiles = 8 whole rounds + 16.  That remainder (tiles mod CUs, when it is at most half the CUs and
// the contraction is >= 16 stages deep) runs split-K — `workspace` >= tn_gemm_grouped_workspace_bytes(...) — so it costs
// a fraction of a round instead of a whole one.  c_f32: fp32 outputs (ldc in floats; both operands contraction-major).
long long tn_gemm_grouped_workspace_bytes(const int* M, const int* N, const int* K, int ngrp) {
  using namespace tn::gemm;
  if (ngrp < 1 || ngrp > MAXSEG) return -1;
  long long tiles = 0;
  int min_stages = 0x7fffffff;
  for (int g = 0; g < ngrp; ++g) {
    tiles += (long long)((M[g] + BM - 1) / BM) * ((N[g] + BN - 1) / BN);
    min_stages = min(min_stages, (K[g] + 63) / 64);
  }
  const int ncu = num_cus();
  const int r = (int)(tiles % ncu);
  if (tiles <= ncu || r == 0 || r * 2 > ncu || min_stages < 16) return 0;
  const int S = min(ncu / r, min_stages / 8);
  return S > 1 ? (long long)S * r * BM * BN * (long long)sizeof(float) : 0;
}

int tn_gemm_bf16_grouped(const void* const* A, const void* const* B, const long long* lda, const long long* ldb,
                         const int* K, void* const* C, const long long* ldc, const int* M, const int* N, int ngrp,
                         int a_kmaj, int b_kmaj, int accumulate, int c_f32, void* workspace, long long workspace_bytes,
                         void* stream) {
  using namespace tn::gemm;
  if (ngrp < 1 || ngrp > MAXSEG) return TN_EINVAL;
  if (!(a_kmaj && b_kmaj)) return TN_EINVAL;               // (weight-gradient mode: the one caller)
  if (TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr) return TN_EINVAL;
  Params p;
  clear_params(p);
  int total = 0, min_stages = 0x7fffffff;
  for (int g = 0; g < ngrp; ++g) {
    const int k = K[g], m = M[g], n = N[g];
    if (m <= 0 || n <= 0 || k <= 0 || (n % 8) || (m % 8) || (lda[g] % 8) || (ldb[g] % 8) || (ldc[g] % (c_f32 ? 4 : 8)))
      return TN_EINVAL;
    if (lda[g] < m || ldb[g] <= 0 || ldc[g] < n) return TN_EINVAL;
    if (((uintptr_t)A[g] | (uintptr_t)B[g] | (uintptr_t)C[g]) & 15) return TN_EINVAL;
    if (((long long)(k - 1) * lda[g] + m) * 2 >= 0x7fffffffLL || ((long long)(k - 1) * ldb[g] + n) * 2 >= 0x7fffffffLL)
      return TN_EINVAL;
    p.seg[g].A = (const tn::bf16_t*)A[g];
    p.seg[g].B = (const tn::bf16_t*)B[g];
    p.seg[g].lda = lda[g];
    p.seg[g].ldb = ldb[g];
    p.seg[g].K = k;
    p.grp[g].C = (tn::bf16_t*)C[g];
    p.grp[g].ldc = ldc[g];
    p.grp[g].M = m;
    p.grp[g].N = n;
    p.grp[g].nbm = (m + BM - 1) / BM;
    p.grp[g].nbn = (n + BN - 1) / BN;
    p.grp[g].start = total;
    p.grp[g].stages = (k + 63) / 64;
    min_stages = min(min_stages, p.grp[g].stages);
    total += p.grp[g].nbm * p.grp[g].nbn;
  }
  for (int g = ngrp; g < MAXSEG; ++g) {
    p.seg[g] = p.seg[0];
    p.grp[g] = p.grp[0];
    p.grp[g].start = 0x7fffffff;
  }
  p.ngrp = ngrp;
  p.accumulate = accumulate;
  p.c_f32 = c_f32;
  const int ncu = num_cus();
  const bool persist = persistent_now();
  hipStream_t st = (hipStream_t)stream;
  // the remainder of the tile list, split-K
  int r = total % ncu, S = 1;
  if (total > ncu && r > 0 && r * 2 <= ncu && min_stages >= 16) S = min(ncu / r, min_stages / 8);
  if (S > 1 && (workspace == nullptr || ((uintptr_t)workspace & 15) ||
                workspace_bytes < (long long)S * r * BM * BN * (long long)sizeof(float)))
    S = 1;                                                  // (no workspace: the remainder runs as whole tiles)
  const int main_tiles = S > 1 ? total - r : total;
  p.tile0 = 0;
  p.ntiles = main_tiles;
  {
    const dim3 grid(persist && main_tiles > ncu ? ncu : main_tiles);
    if (c_f32)
      hipLaunchKernelGGL((gemm_kernel<true, true, kDPL, kDAS, kDIL, false, false, true, EPI_GROUPED>), grid, dim3(NT), 0, st, p);
    else
      hipLaunchKernelGGL((gemm_kernel<true, true, kDPL, kDAS, kDIL, false, false, false, EPI_GROUPED>), grid, dim3(NT), 0, st, p);
  }
  if (S > 1) {
    float* ws = (float*)workspace;
    for (int g = 0; g < ngrp; ++g) {
      const int end = p.grp[g].start + p.grp[g].nbm * p.grp[g].nbn;
      const int lo = max(main_tiles, p.grp[g].start), hi = end;
      if (lo >= hi) continue;
      Params q;
      clear_params(q);
      q.seg[0] = p.seg[g];
      q.seg[1] = q.seg[2] = q.seg[0];
      q.M = p.grp[g].M;
      q.N = p.grp[g].N;
      q.C = p.grp[g].C;
      q.ldc = p.grp[g].ldc;
      q.accumulate = accumulate;
      q.nbm = p.grp[g].nbm;
      q.nbn = p.grp[g].nbn;
      q.stages = p.grp[g].stages;
      q.splitk = S;
      q.kchunk = (q.stages + S - 1) / S;
      q.ws = ws;
      q.tile0 = lo - p.grp[g].start;
      q.ntiles = hi - lo;
      q.c_f32 = c_f32;
      const int units = q.ntiles * S;
      const dim3 grid(persist && units > ncu ? ncu : units);
      hipLaunchKernelGGL((gemm_kernel<true, true, kDPL, kDAS, kDIL, false, true>), grid, dim3(NT), 0, st, q);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)q.ntiles * 32u), dim3(256), 0, st, q.ws, q.splitk, q.M, q.N,
                         q.nbm, q.nbn, q.tile0, q.ntiles, q.C, q.ldc, (const tn::bf16_t*)nullptr, q.accumulate, q.c_f32,
                         (const float*)nullptr, (tn::bf16_t*)nullptr);
      ws += (long long)S * q.ntiles * BM * BN;
    }
  }
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// The MLP's gate and up products with the SwiGLU fused into the epilogue (transformers' LlamaMLP.forward; liger's fused
// SwiGLU at touchnet/models/llama/__init__.py:11-15): gate[M, I] = x Wg^T, up[M, I] = x Wu^T, act = silu(gate) * up, all
// bf16 with row pitch ldc; x [M, K] (pitch ldx), Wg / Wu [I, K] (pitch ldw).  -22 unless K % 64 == 0, I % 8 == 0, pitches % 8.
int tn_gemm_bf16_swiglu_fwd(const void* x, const void* wg, const void* wu, void* gate, void* up, void* act, int M, int I,
                            int K, long long ldx, long long ldw, long long ldc, void* stream) {
  using namespace tn::gemm;
  if (M <= 0 || I <= 0 || K <= 0 || (K % 64) || (I % 8) || (ldx % 8) || (ldw % 8) || (ldc % 8) || ldc < I) return TN_EINVAL;
  if (ldx < K || ldw < K) return TN_EINVAL;        // (row pitches of the row-stored operands: at least the contraction)
  if (((uintptr_t)x | (uintptr_t)wg | (uintptr_t)wu | (uintptr_t)gate | (uintptr_t)up | (uintptr_t)act) & 15) return TN_EINVAL;
  if ((long long)288 * ldx * 2 >= 0x7fffffffLL || (long long)288 * ldw * 2 >= 0x7fffffffLL) return TN_EINVAL;
  if (TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr) return TN_EINVAL;
  Params p;
  clear_params(p);
  p.seg[0].A = p.seg[1].A = (const tn::bf16_t*)x;
  p.seg[0].B = (const tn::bf16_t*)wg;
  p.seg[1].B = (const tn::bf16_t*)wu;
  p.seg[0].lda = p.seg[1].lda = ldx;
  p.seg[0].ldb = p.seg[1].ldb = ldw;
  p.seg[0].K = p.seg[1].K = K;
  p.seg[2] = p.seg[0];
  p.stages = K / 64;
  p.M = M;
  p.N = I;
  p.C = (tn::bf16_t*)gate;
  p.C2 = (tn::bf16_t*)up;
  p.C3 = (tn::bf16_t*)act;
  p.ldc = ldc;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (I + 127) / 128;
  p.ntiles = p.nbm * p.nbn;
  const int ncu = num_cus();
  const dim3 grid(persistent_now() && p.ntiles > ncu ? ncu : p.ntiles);
  if (use_m16())
    hipLaunchKernelGGL((gemm16_kernel<false, false, kDPL, EPI_SWIGLU_FWD>), grid, dim3(NT), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<false, false, kDPL, kDAS, kDIL, false, false, false, EPI_SWIGLU_FWD>), grid, dim3(NT), 0,
                       (hipStream_t)stream, p);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// Its backward counterpart: d(act) = dY W_down (dY [M, H] pitch lddy, W_down [H, I] read contraction-major, pitch ldw) is
// formed in the accumulators only; the epilogue reads gate / up [M, I] (pitch ld) and writes d(gate), d(up) (pitch ld).
// -22 unless H % 64 == 0, I % 8 == 0, pitches % 8.
int tn_gemm_bf16_swiglu_bwd(const void* dy, const void* wd, const void* gate, const void* up, void* dgate, void* dup,
                            int M, int I, int H, long long lddy, long long ldw, long long ld, void* stream) {
  using namespace tn::gemm;
  if (M <= 0 || I <= 0 || H <= 0 || (H % 64) || (I % 8) || (lddy % 8) || (ldw % 8) || (ld % 8) || ld < I) return TN_EINVAL;
  if (lddy < H) return TN_EINVAL;                  // (dY is row-stored: its pitch is at least the contraction)
  if (((uintptr_t)dy | (uintptr_t)wd | (uintptr_t)gate | (uintptr_t)up | (uintptr_t)dgate | (uintptr_t)dup) & 15)
    return TN_EINVAL;
  if ((long long)288 * lddy * 2 >= 0x7fffffffLL || ldw < I || ((long long)(H - 1) * ldw + I) * 2 >= 0x7fffffffLL)
    return TN_EINVAL;
  if (TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr) return TN_EINVAL;
  Params p;
  clear_params(p);
  p.seg[0].A = (const tn::bf16_t*)dy;
  p.seg[0].B = (const tn::bf16_t*)wd;
  p.seg[0].lda = lddy;
  p.seg[0].ldb = ldw;
  p.seg[0].K = H;
  p.seg[1] = p.seg[2] = p.seg[0];
  p.stages = H / 64;
  p.M = M;
  p.N = I;
  p.C = (tn::bf16_t*)dgate;
  p.C2 = (tn::bf16_t*)dup;
  p.E1 = (const tn::bf16_t*)gate;
  p.E2 = (const tn::bf16_t*)up;
  p.ldc = p.lde = ld;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (I + BN - 1) / BN;
  p.ntiles = p.nbm * p.nbn;
  const int ncu = num_cus();
  const dim3 grid(persistent_now() && p.ntiles > ncu ? ncu : p.ntiles);
  if (use_m16())
    hipLaunchKernelGGL((gemm16_kernel<false, true, kDPL, EPI_SWIGLU_BWD>), grid, dim3(NT), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<false, true, kDPL, kDAS, kDIL, false, false, false, EPI_SWIGLU_BWD>), grid, dim3(NT), 0,
                       (hipStream_t)stream, p);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// Weight gradient AND bias gradient of a linear layer y = x W^T + b in one launch: C[M, N] (bf16, or float when c_f32;
// += when accumulate) = A[K, M]^T . B[K, N] and bias_grad[M] (bf16) = column sums of A — A = dY [tokens, M], B = x [tokens, N]
// as stored.  The sums come from the A fragments the matrix pipe reads anyway (EPI_BIASG above): no separate pass over dY.
// splitk >= 2: split-K through `workspace` (>= (splitk * tiles * 65536 + splitk * ceil(M / 256) * 256) * 4 bytes).
int tn_gemm_bf16_wgrad_bias(const void* A, const void* B, long long lda, long long ldb, int K, void* C, void* bias_grad,
                            int M, int N, long long ldc, int accumulate, int c_f32, int splitk, void* workspace,
                            long long workspace_bytes, void* stream) {
  if (bias_grad == nullptr) return TN_EINVAL;
  return gemm_launch(&A, &B, &lda, &ldb, &K, 1, 1, 1, C, nullptr, nullptr, M, N, ldc, 0, accumulate,
                     splitk > 1 ? splitk : 1, 0, workspace, workspace_bytes, stream, c_f32, bias_grad);
}

// A q / k projection with the rotary embedding in the epilogue: out[M, N] = rope(x W^T + bias), N = heads x head_dim
// (head_dim 64 or 128; N % 256 == 0 for 128, N % 64 == 0 for 64), cos / sin [M, head_dim / 2] bf16 = one table row per
// output row (tn_rope_table).  -22 unless K % 64 == 0 and the pitches are multiples of 8.
int tn_gemm_bf16_rope(const void* x, const void* w, const void* bias, const void* cos_t, const void* sin_t, void* out, int M,
                      int N, int K, long long ldx, long long ldw, long long ldc, int head_dim, void* stream) {
  using namespace tn::gemm;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (ldx % 8) || (ldw % 8) || (ldc % 8) || ldc < N) return TN_EINVAL;
  if (ldx < K || ldw < K) return TN_EINVAL;        // (row pitches of the row-stored operands: at least the contraction)
  if (!((head_dim == 128 && N % 256 == 0) || (head_dim == 64 && N % 64 == 0))) return TN_EINVAL;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) return TN_EINVAL;
  if (((uintptr_t)cos_t | (uintptr_t)sin_t | (uintptr_t)bias) & 7) return TN_EINVAL;
  if (cos_t == nullptr || sin_t == nullptr) return TN_EINVAL;
  if ((long long)288 * ldx * 2 >= 0x7fffffffLL || (long long)288 * ldw * 2 >= 0x7fffffffLL) return TN_EINVAL;
  if (TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr) return TN_EINVAL;
  Params p;
  clear_params(p);
  p.seg[0].A = (const tn::bf16_t*)x;
  p.seg[0].B = (const tn::bf16_t*)w;
  p.seg[0].lda = ldx;
  p.seg[0].ldb = ldw;
  p.seg[0].K = K;
  p.seg[1] = p.seg[2] = p.seg[0];
  p.stages = K / 64;
  p.M = M;
  p.N = N;
  p.C = (tn::bf16_t*)out;
  p.bias = (const tn::bf16_t*)bias;
  p.ldc = ldc;
  p.rope_cos = (const tn::bf16_t*)cos_t;
  p.rope_sin = (const tn::bf16_t*)sin_t;
  p.rope_d = head_dim;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (N + BN - 1) / BN;
  p.ntiles = p.nbm * p.nbn;
  const int ncu = num_cus();
  const dim3 grid(persistent_now() && p.ntiles > ncu ? ncu : p.ntiles);
  if (use_m16())
    hipLaunchKernelGGL((gemm16_kernel<false, false, kDPL, EPI_ROPE>), grid, dim3(NT), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<false, false, kDPL, kDAS, kDIL, false, false, false, EPI_ROPE>), grid, dim3(NT), 0,
                       (hipStream_t)stream, p);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// The audio tower's MLP, first half: pre[M, N] = x[M, K] W[N, K]^T + bias AND act = gelu(pre) from one launch (exact-erf
// GELU, common.h gelu_f: the bits of tn_gelu_fwd on the rounded pre).  -22 unless K % 64 == 0, N % 8 == 0, pitches % 8 and
// >= the rows they span, 16-byte aligned bases.
int tn_gemm_bf16_gelu_fwd(const void* x, const void* w, const void* bias, void* pre, void* act, int M, int N, int K,
                          long long ldx, long long ldw, long long ldc, void* stream) {
  using namespace tn::gemm;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 8) || (ldx % 8) || (ldw % 8) || (ldc % 8) || ldc < N) return TN_EINVAL;
  if (ldx < K || ldw < K) return TN_EINVAL;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)pre | (uintptr_t)act) & 15) return TN_EINVAL;
  if (((uintptr_t)bias & 7) || pre == act) return TN_EINVAL;
  if ((long long)288 * ldx * 2 >= 0x7fffffffLL || (long long)288 * ldw * 2 >= 0x7fffffffLL) return TN_EINVAL;
  if (TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr || !use_m16()) return TN_EINVAL;
  Params p;
  clear_params(p);
  p.seg[0].A = (const tn::bf16_t*)x;
  p.seg[0].B = (const tn::bf16_t*)w;
  p.seg[0].lda = ldx;
  p.seg[0].ldb = ldw;
  p.seg[0].K = K;
  p.seg[1] = p.seg[2] = p.seg[0];
  p.stages = K / 64;
  p.M = M;
  p.N = N;
  p.C = (tn::bf16_t*)pre;
  p.C2 = (tn::bf16_t*)act;
  p.bias = (const tn::bf16_t*)bias;
  p.ldc = ldc;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (N + BN - 1) / BN;
  p.ntiles = p.nbm * p.nbn;
  const int ncu = num_cus();
  const dim3 grid(persistent_now() && p.ntiles > ncu ? ncu : p.ntiles);
  hipLaunchKernelGGL((gemm16_kernel<false, false, kDPL, EPI_GELU_FWD>), grid, dim3(NT), 0, (hipStream_t)stream, p);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// ... and the backward of its second half: d(pre)[M, I] = (dY[M, H] W2[H, I]) o gelu'(pre) — the product d(act) is formed in
// the accumulators only (W2 read contraction-major, pitch ldw; pre and d(pre) with pitch ld).  -22 unless H % 64 == 0,
// I % 8 == 0, pitches % 8.
int tn_gemm_bf16_gelu_bwd(const void* dy, const void* w2, const void* pre, void* dpre, int M, int I, int H, long long lddy,
                          long long ldw, long long ld, void* stream) {
  using namespace tn::gemm;
  if (M <= 0 || I <= 0 || H <= 0 || (H % 64) || (I % 8) || (lddy % 8) || (ldw % 8) || (ld % 8) || ld < I) return TN_EINVAL;
  if (lddy < H) return TN_EINVAL;
  if (((uintptr_t)dy | (uintptr_t)w2 | (uintptr_t)pre | (uintptr_t)dpre) & 15) return TN_EINVAL;
  if ((long long)288 * lddy * 2 >= 0x7fffffffLL || ldw < I || ((long long)(H - 1) * ldw + I) * 2 >= 0x7fffffffLL)
    return TN_EINVAL;
  if (TN_GEMM_DEFAULT_VARIANT >= 1000 || getenv("TN_GEMM_VARIANT") != nullptr || !use_m16()) return TN_EINVAL;
  Params p;
  clear_params(p);
  p.seg[0].A = (const tn::bf16_t*)dy;
  p.seg[0].B = (const tn::bf16_t*)w2;
  p.seg[0].lda = lddy;
  p.seg[0].ldb = ldw;
  p.seg[0].K = H;
  p.seg[1] = p.seg[2] = p.seg[0];
  p.stages = H / 64;
  p.M = M;
  p.N = I;
  p.C = (tn::bf16_t*)dpre;
  p.E1 = (const tn::bf16_t*)pre;
  p.ldc = p.lde = ld;
  p.nbm = (M + BM - 1) / BM;
  p.nbn = (I + BN - 1) / BN;
  p.ntiles = p.nbm * p.nbn;
  const int ncu = num_cus();
  const dim3 grid(persistent_now() && p.ntiles > ncu ? ncu : p.ntiles);
  hipLaunchKernelGGL((gemm16_kernel<false, true, kDPL, EPI_GELU_BWD>), grid, dim3(NT), 0, (hipStream_t)stream, p);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// C[M,N] = A[M,K] · B[N,K]^T (+ bias) (+ C if accumulate); optional transposed copy Ct[N,M]: the round-2 entry point,
// kept as the single-segment, both-operands-contraction-contiguous case of tn_gemm_bf16.
int tn_gemm_bf16_tn(const void* A, const void* B, void* C, void* Ct, const void* bias, int M, int N, int K,
                    long long lda, long long ldb, long long ldc, long long ldct, int accumulate, void* stream) {
  return tn_gemm_bf16(&A, &B, &lda, &ldb, &K, 1, 0, 0, C, Ct, bias, M, N, ldc, ldct, accumulate, stream);
}

}  // extern "C"
