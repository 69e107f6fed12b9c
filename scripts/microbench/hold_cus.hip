// Stand-in for a collective kernel that HOLDS compute units (RCCL's reduce-scatter / all-gather kernels: a few dozen
// long-running workgroups): `n` workgroups, each taking a whole CU's LDS so that no two share one, spin for `us`
// microseconds.  scripts/persist_vs_sidecopy.py launches it on a side stream beside a loop of GEMMs.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(1024) void hold_kernel(long long ticks, int* sink) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < ticks) t = wall_clock64();
  if (threadIdx.x == 0 && sink != nullptr && t == 0) *sink = lds[0];
}
extern "C" int hold_cus(int n, double us, void* stream) {
  int rate = 100000;                                        // wall_clock64 ticks at 100 MHz on gfx9
  (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);      // kHz
  (void)hipFuncSetAttribute((const void*)hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipLaunchKernelGGL(hold_kernel, dim3(n), dim3(1024), 96 * 1024, (hipStream_t)stream, (long long)(us * rate / 1000.0), nullptr);
  return (int)hipGetLastError();
}
