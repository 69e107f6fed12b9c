// What does ds_read_b64_tr_b16 (gfx950) deliver?  LDS holds element index = 64 * row + col of a [64][64] 16-bit tile;
// every lane supplies the address of 4 consecutive elements (8 bytes); prints, per lane, the (row, col) of the four
// elements it received.  Build + run on the GPU box: hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;

__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) sm[i] = i;
  __syncthreads();
  const int l = threadIdx.x;
  int row, col;
  if (mode == 0) { row = l & 15; col = (l >> 4) * 4; }          // 16 rows x 4 column groups
  else if (mode == 1) { row = l >> 2; col = (l & 3) * 4; }      // 16 rows x 4 groups, lane-minor groups
  else { row = l & 31; col = (l >> 5) * 4; }                    // 32 rows x 2 groups
  const uint16_t* p = sm + row * 64 + col;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (lane: 4 x (row,col) received)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int e = 0; e < 4; ++e) printf(" (%2d,%2d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
      printf("\n");
    }
  }
  return 0;
}
