// Helpers shared by the "stream" attention kernels (attn_fwd_stream.hip, attn_bwd_dq_stream.hip): inline-asm LDS reads with
// hand-counted waits (hipcc serialises compiler-visible reads of an LDS-DMA ring — attn_common.h — and pairs every MFMA of
// a chain with its own read), compile-time loops, half-wave exchanges.
#pragma once
#include <utility>

#include "attn_common.h"

namespace tn {

namespace fstream {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

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

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// v_permlane32_swap: {a with its upper half replaced by b's lower half, b with its lower half replaced by a's upper half}
__device__ __forceinline__ u32x2_t swap32(uint32_t a, uint32_t b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return u32x2_t{r[0], r[1]};
}
// sum / max over the two 32-lane halves of a wave
__device__ __forceinline__ float half_sum(float x) {
  const u32x2_t r = swap32(__float_as_uint(x), __float_as_uint(x));
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float half_max(float x) {
  const u32x2_t r = swap32(__float_as_uint(x), __float_as_uint(x));
  return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}

}  // namespace fstream

}  // namespace tn
