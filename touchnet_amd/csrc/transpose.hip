// bf16 matrix transpose, dst[c][r] = src[r][c], for the weight-gradient GEMMs of the linear layers.
//
// Why it exists: autograd's dW = dY^T X contracts over the token dimension, which is the SLOW dimension of both
// operands ("NT").  hipBLASLt runs that layout at ~0.95-1.0 PFLOP/s on MI355X, but the same product with both
// operands contraction-contiguous ("TN", the layout of the forward GEMM) at 1.4-1.55 PFLOP/s
// (scripts/wgrad_layout_bench.py).  Transposing dY and X costs 4 bytes of HBM traffic per element at ~5 TB/s,
// a fraction of the GEMM time saved.  Replaces the implicit transposes inside
// torch.nn.functional.linear's backward (reference: every nn.Linear of touchnet/models/*, e.g. the HF Llama
// blocks driven by touchnet/bin/train.py:440-470).
//
// HBM-bound: no LDS.  A thread owns an 8 x 8 block: 8 row loads of 16 B (8 adjacent lanes = one 128-B line of a
// source row), a 32-instruction v_perm register transpose, 8 column stores of 16 B (8 adjacent row-blocks = one
// 128-B line of a destination row).
#include "common.h"

namespace tn {

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int rows, int cols, long long src_ld, long long dst_ld) {
  // workgroup tile: 256 rows x 64 columns; lane = 8 column chunks (fast) x 8 row blocks, 4 waves stack row blocks
  const int tid = threadIdx.x;
  const int cx = tid & 7, rb = tid >> 3;
  const int c0 = (blockIdx.x * 8 + cx) * 8;
  const int r0 = (blockIdx.y * 32 + rb) * 8;
  if (c0 >= cols || r0 >= rows) return;
  uint4 in[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) in[i] = *reinterpret_cast<const uint4*>(src + (long long)(r0 + i) * src_ld + c0);
  const uint32_t w[8][4] = {{in[0].x, in[0].y, in[0].z, in[0].w}, {in[1].x, in[1].y, in[1].z, in[1].w},
                            {in[2].x, in[2].y, in[2].z, in[2].w}, {in[3].x, in[3].y, in[3].z, in[3].w},
                            {in[4].x, in[4].y, in[4].z, in[4].w}, {in[5].x, in[5].y, in[5].z, in[5].w},
                            {in[6].x, in[6].y, in[6].z, in[6].w}, {in[7].x, in[7].y, in[7].z, in[7].w}};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int wi = c >> 1;
    const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;    // high / low halves of the two dwords
    uint4 o;
    o.x = __builtin_amdgcn_perm(w[1][wi], w[0][wi], sel);
    o.y = __builtin_amdgcn_perm(w[3][wi], w[2][wi], sel);
    o.z = __builtin_amdgcn_perm(w[5][wi], w[4][wi], sel);
    o.w = __builtin_amdgcn_perm(w[7][wi], w[6][wi], sel);
    *reinterpret_cast<uint4*>(dst + (long long)(c0 + c) * dst_ld + r0) = o;
  }
}

}  // namespace tn

extern "C" int tn_transpose_bf16(const void* src, void* dst, int rows, int cols, long long src_ld, long long dst_ld,
                                 void* stream) {
  if (rows <= 0 || cols <= 0) return TN_OK;
  // 16-byte accesses on both sides: 8-element granularity of shapes, leading dimensions and base addresses
  if ((rows | cols) & 7 || (src_ld | dst_ld) & 7 || src_ld < cols || dst_ld < rows) return TN_EINVAL;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return TN_EINVAL;
  dim3 grid((cols + 63) / 64, (rows + 255) / 256), block(256);
  hipLaunchKernelGGL(tn::transpose_bf16_kernel, grid, block, 0, (hipStream_t)stream, (const tn::bf16_t*)src,
                     (tn::bf16_t*)dst, rows, cols, src_ld, dst_ld);
  TN_LAUNCH_CHECK();
  return TN_OK;
}
