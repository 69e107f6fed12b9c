// Packed, per-sentence-normalised cross-entropy over the vocabulary (+ accuracy), fwd and bwd.
//
// Replaces  touchnet/loss/__init__.py:7-28          (_cross_entropy_loss: upcast + F.cross_entropy 'none')
//           touchnet/loss/cross_entropy.py:34-49     (per-token / per-sentence / per-global-batch reductions)
//           touchnet/utils/metrics.py:41-50          (argmax accuracy)
//
// Forward  (one 256-thread block per packed position n, logits row [V] read ONCE from HBM):
//   online max / sum-exp / argmax in registers, 16-byte loads;  rows with label == ignore_index are
//   not read at all (nll = 0) — on ASR-SFT batches > 90 % of the positions are audio/prompt slots.
//   Outputs per row: nll[n], lse[n] (natural log), hit[n] = (argmax == label).
// Reduce   (one block, no host sync — the reference does two .item() syncs here):
//   loss_per_token  = sum(nll)/n_valid            (0 when sum <= 1e-6 or n_valid == 0)
//   loss_per_sample = sum_n nll[n]/sentence_lens[n] / num_sentence
//   accuracy        = sum(hit)/n_valid            (0 when n_valid == 0)
// Backward (row block again): dlogits[n,v] = (exp(x - lse) - [v == label]) * g / (sentence_lens[n] * num_sentence)
//   g = upstream gradient of loss_per_sample (device scalar); ignored rows get zeros.  May run in place.
// Algorithmic bytes: fwd V*sizeof(T) per valid row; bwd 2*V*sizeof(T) per valid row (+V*sizeof(T) zero-fill
// per ignored row when not in place).
#include "common.h"

namespace tn {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct RowStat {
  float m;    // running max
  float s;    // running sum of exp(x - m)
  float bv;   // best value
  int bi;     // best index (first occurrence)
};

__device__ __forceinline__ void stat_push(RowStat& st, float x, int idx) {
  if (x > st.bv) {  // strict: keeps the first index among equal values within a thread's ascending scan
    st.bv = x;
    st.bi = idx;
  }
  if (x > st.m) {
    st.s = st.s * exp2f((st.m - x) * kLog2e) + 1.f;
    st.m = x;
  } else {
    st.s += exp2f((x - st.m) * kLog2e);
  }
}

__device__ __forceinline__ void stat_merge(RowStat& a, const RowStat& b) {
  if (b.bv > a.bv || (b.bv == a.bv && b.bi < a.bi)) {
    a.bv = b.bv;
    a.bi = b.bi;
  }
  const float m = fmaxf(a.m, b.m);
  if (m == -INFINITY) return;
  a.s = a.s * exp2f((a.m - m) * kLog2e) + b.s * exp2f((b.m - m) * kLog2e);
  a.m = m;
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     float* __restrict__ nll, float* __restrict__ lse,
                                                     int* __restrict__ hit, int V, int64_t ignore_index) {
  constexpr int N = Vec16<T>::N;
  __shared__ RowStat sm[4];
  const int n = blockIdx.x;
  const int64_t lab = labels[n];
  if (lab == ignore_index) {
    if (threadIdx.x == 0) {
      nll[n] = 0.f;
      lse[n] = 0.f;
      hit[n] = 0;
    }
    return;
  }
  const T* row = logits + (size_t)n * V;
  RowStat st = {-INFINITY, 0.f, -INFINITY, 0x7fffffff};
  const bool vec_ok = (V % N == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0);
  if (vec_ok) {
    const int nvec = V / N;
    for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
      Vec16<T> a;
      float f[N];
      a.load(row + (size_t)v * N);
      a.unpack(f);
#pragma unroll
      for (int j = 0; j < N; ++j) stat_push(st, f[j], v * N + j);
    }
  } else {
    for (int i = threadIdx.x; i < V; i += blockDim.x) stat_push(st, Elem<T>::ld(row + i), i);
  }
  // wave merge, then block merge
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    RowStat other;
    other.m = __shfl_xor(st.m, o, 64);
    other.s = __shfl_xor(st.s, o, 64);
    other.bv = __shfl_xor(st.bv, o, 64);
    other.bi = __shfl_xor(st.bi, o, 64);
    stat_merge(st, other);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = st;
  __syncthreads();
  if (threadIdx.x == 0) {
    RowStat r = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) stat_merge(r, sm[i]);
    const float l = r.m + logf(r.s);
    const float xl = (lab >= 0 && lab < V) ? Elem<T>::ld(row + lab) : 0.f;
    nll[n] = l - xl;
    lse[n] = l;
    hit[n] = (r.bi == (int)lab) ? 1 : 0;
  }
}

// out[0]=loss_per_sample out[1]=loss_per_token out[2]=accuracy out[3]=n_valid (float)
__global__ __launch_bounds__(1024) void ce_reduce_kernel(const float* __restrict__ nll, const int* __restrict__ hit,
                                                         const int64_t* __restrict__ labels,
                                                         const int64_t* __restrict__ sentence_lens,
                                                         const float* __restrict__ num_sentence,
                                                         float* __restrict__ out, int n, int64_t ignore_index) {
  __shared__ float sm[16];
  float s_tok = 0.f, s_sent = 0.f, s_hit = 0.f, s_cnt = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (labels[i] != ignore_index) {
      const float v = nll[i];
      s_tok += v;
      s_sent += v / (float)sentence_lens[i];
      s_hit += (float)hit[i];
      s_cnt += 1.f;
    }
  }
  s_tok = block_sum(s_tok, sm);
  s_sent = block_sum(s_sent, sm);
  s_hit = block_sum(s_hit, sm);
  s_cnt = block_sum(s_cnt, sm);
  if (threadIdx.x == 0) {
    out[0] = s_sent / num_sentence[0];
    out[1] = (s_tok > 1e-6f && s_cnt > 0.f) ? s_tok / s_cnt : 0.f;
    out[2] = s_cnt > 0.f ? s_hit / s_cnt : 0.f;
    out[3] = s_cnt;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, T* __restrict__ dlogits,
                                                     const int64_t* __restrict__ labels,
                                                     const int64_t* __restrict__ sentence_lens,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ num_sentence,
                                                     const float* __restrict__ grad_out, int V,
                                                     int64_t ignore_index) {
  constexpr int N = Vec16<T>::N;
  const int n = blockIdx.x;
  const int64_t lab = labels[n];
  const T* row = logits + (size_t)n * V;
  T* drow = dlogits + (size_t)n * V;
  const bool vec_ok = (V % N == 0) && ((reinterpret_cast<uintptr_t>(row) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(drow) & 15) == 0);
  if (lab == ignore_index) {
    if (vec_ok) {
      Vec16<T> z;
      float zf[N];
#pragma unroll
      for (int j = 0; j < N; ++j) zf[j] = 0.f;
      z.pack(zf);
      for (int v = threadIdx.x; v < V / N; v += blockDim.x) z.store(drow + (size_t)v * N);
    } else {
      for (int i = threadIdx.x; i < V; i += blockDim.x) Elem<T>::st(drow + i, 0.f);
    }
    return;
  }
  const float scale = grad_out[0] / ((float)sentence_lens[n] * num_sentence[0]);
  const float l2 = lse[n] * kLog2e;
  if (vec_ok) {
    for (int v = threadIdx.x; v < V / N; v += blockDim.x) {
      Vec16<T> a;
      float f[N];
      a.load(row + (size_t)v * N);
      a.unpack(f);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float p = exp2f(f[j] * kLog2e - l2);
        if (v * N + j == (int)lab) p -= 1.f;
        f[j] = p * scale;
      }
      a.pack(f);
      a.store(drow + (size_t)v * N);
    }
  } else {
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
      float p = exp2f(Elem<T>::ld(row + i) * kLog2e - l2);
      if (i == (int)lab) p -= 1.f;
      Elem<T>::st(drow + i, p * scale);
    }
  }
}

}  // namespace tn

using namespace tn;

extern "C" {

// logits [n, V] (dtype 0 = fp32, 1 = bf16), labels/sentence_lens int64 [n], num_sentence: device float[1].
// nll/lse float [n], hit int32 [n], out float[4] = {loss_per_sample, loss_per_token, accuracy, n_valid}.
int tn_ce_forward(const void* logits, const long long* labels, const long long* sentence_lens,
                  const float* num_sentence, float* nll, float* lse, int* hit, float* out, int n, int V,
                  long long ignore_index, int dtype, void* stream) {
  if (n <= 0 || V <= 0) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3(n), dim3(256), 0, st, (const float*)logits,
                       (const int64_t*)labels, nll, lse, hit, V, (int64_t)ignore_index);
  } else if (dtype == 1) {
    hipLaunchKernelGGL((ce_fwd_kernel<bf16_t>), dim3(n), dim3(256), 0, st, (const bf16_t*)logits,
                       (const int64_t*)labels, nll, lse, hit, V, (int64_t)ignore_index);
  } else {
    return TN_EINVAL;
  }
  TN_LAUNCH_CHECK();
  if (out) {
    hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, st, nll, hit, (const int64_t*)labels,
                       (const int64_t*)sentence_lens, num_sentence, out, n, (int64_t)ignore_index);
    TN_LAUNCH_CHECK();
  }
  return TN_OK;
}

// Reduction only (used by the chunked fused linear+CE path after all chunks have filled nll/hit).
int tn_ce_reduce(const float* nll, const int* hit, const long long* labels, const long long* sentence_lens,
                 const float* num_sentence, float* out, int n, long long ignore_index, void* stream) {
  if (n <= 0) return TN_EINVAL;
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nll, hit,
                     (const int64_t*)labels, (const int64_t*)sentence_lens, num_sentence, out, n,
                     (int64_t)ignore_index);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

// dlogits may alias logits.  grad_out: device float[1] (upstream d loss_per_sample).
int tn_ce_backward(const void* logits, void* dlogits, const long long* labels, const long long* sentence_lens,
                   const float* lse, const float* num_sentence, const float* grad_out, int n, int V,
                   long long ignore_index, int dtype, void* stream) {
  if (n <= 0 || V <= 0) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(n), dim3(256), 0, st, (const float*)logits, (float*)dlogits,
                       (const int64_t*)labels, (const int64_t*)sentence_lens, lse, num_sentence, grad_out, V,
                       (int64_t)ignore_index);
  } else if (dtype == 1) {
    hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), dim3(n), dim3(256), 0, st, (const bf16_t*)logits,
                       (bf16_t*)dlogits, (const int64_t*)labels, (const int64_t*)sentence_lens, lse, num_sentence,
                       grad_out, V, (int64_t)ignore_index);
  } else {
    return TN_EINVAL;
  }
  TN_LAUNCH_CHECK();
  return TN_OK;
}

}  // extern "C"
