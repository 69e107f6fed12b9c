// VALU issue-cost microbenchmark for gfx950 (kernel development aid, not part of the library).
// Each test runs REP x 64 copies of one instruction (8 independent register chains) in ONE wave per SIMD
// (256-thread block) or TWO (512-thread block) and reports shader cycles (s_memtime) per instruction per wave.
// build: hipcc --offload-arch=gfx950 -O3 scripts/microbench/valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INS)                                                                                               \
  REP8(asm volatile(INS "\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                    : "v"(b0), "v"(b1));)

template <int TEST>
__global__ void bench(unsigned long long* out, float seed, int reps) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = seed * 0.5f, b1 = seed * 0.25f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (TEST == 0) { BODY("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9") }
    if (TEST == 1) { BODY("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7") }
    if (TEST == 2) { BODY("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8") }
    if (TEST == 3) { BODY("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9") }
    if (TEST == 4) { BODY("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8") }
    if (TEST == 5) { BODY("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9") }
    // exp interleaved 1:1 with fma (does the transcendental unit overlap plain VALU?)
    if (TEST == 6) { BODY("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %8, %9") }
    // exp : plain = 1 : 3
    if (TEST == 7) { BODY("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %9\n v_add_f32 %2, %2, %8\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %8, %9\n v_add_f32 %6, %6, %8\n v_fma_f32 %7, %7, %8, %9") }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 0;
}

// packed f32 ops need register pairs: separate kernel
template <int TEST>
__global__ void bench_pk(unsigned long long* out, float seed, int reps) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  f2 b0 = {seed * 0.5f, seed * 0.3f}, b1 = {seed * 0.25f, seed * 0.2f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (TEST == 0) { BODY("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9") }
    if (TEST == 1) { BODY("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8") }
    if (TEST == 2) { BODY("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8") }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s.x + s.y == 12345.678f) out[0] = 0;
}

template <typename K>
static void run(const char* name, K kern, int threads) {
  unsigned long long* d;
  hipMalloc(&d, 8 * 8 * sizeof(unsigned long long));
  const int reps = 64;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, 1.0f, reps);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double n = reps * 64.0;
  printf("%-34s %d waves/SIMD: %6.2f cycles/instr/wave  (SIMD throughput %.2f cycles/instr)\n", name, threads / 256,
         h[0] / n, h[0] / n / (threads / 256));
  hipFree(d);
}

int main() {
  for (int threads : {256, 512}) {
    run("v_fma_f32", bench<0>, threads);
    run("v_exp_f32", bench<1>, threads);
    run("v_add_f32", bench<2>, threads);
    run("v_max3_f32", bench<3>, threads);
    run("v_cvt_pk_bf16_f32", bench<4>, threads);
    run("v_perm_b32", bench<5>, threads);
    run("exp:fma 1:1", bench<6>, threads);
    run("exp:plain 1:3", bench<7>, threads);
    run("v_pk_fma_f32", bench_pk<0>, threads);
    run("v_pk_add_f32", bench_pk<1>, threads);
    run("v_pk_mul_f32", bench_pk<2>, threads);
  }
  return 0;
}
