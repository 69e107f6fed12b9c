// Round 5 probe: sustained bf16 MFMA rate of the two dense shapes under the chip's power cap, operands in registers only
// (no LDS, no memory): is v_mfma_f32_16x16x32_bf16 (what hipBLASLt's gfx950 kernels issue) cheaper per flop than
// v_mfma_f32_32x32x16_bf16 (what csrc/gemm.hip issues)?   hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void burn(float* out, int iters, unsigned seed) {
  // operands with realistic bit activity (pseudo-random bf16 values), different per lane
  unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 97u;
  bf16x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) {
      s = s * 1664525u + 1013904223u;
      a[i][e] = (__bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f);
      s = s * 1664525u + 1013904223u;
      b[i][e] = (__bf16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f);
    }
  if constexpr (SHAPE == 32) {
    f32x16_t acc[8];
    for (int i = 0; i < 8; ++i)
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i)
      for (int e = 0; e < 16; ++e) t += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  } else {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i)
      for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i)
      for (int e = 0; e < 4; ++e) t += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  }
}

template <int SHAPE>
double run(float* out, int blocks, int iters, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(burn<SHAPE>, dim3(blocks), dim3(512), 0, 0, out, iters, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(burn<SHAPE>, dim3(blocks), dim3(512), 0, 0, out, iters, 2u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)reps * blocks * 8 /*waves*/ * iters * 8 * 32768.0;   // per iteration and wave: 8 x 32x32x16 = 16 x 16x16x32
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 512 * sizeof(float));
  const int blocks = 256, iters = 200000;    // one 8-wave workgroup per CU (two waves per SIMD, like the GEMM), ~1 s per launch
  for (int round = 0; round < 3; ++round) {
    printf("round %d: 32x32x16 %.1f TFLOP/s", round, run<32>(out, blocks, iters, 3));
    printf("   16x16x32 %.1f TFLOP/s\n", run<16>(out, blocks, iters, 3));
  }
  return 0;
}
