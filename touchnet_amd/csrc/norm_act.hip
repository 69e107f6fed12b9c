// HBM-bound row kernels: (add+)RMSNorm, (add+)LayerNorm, SwiGLU, GELU, RoPE.
//
// Reference arithmetic being replaced (all executed as un-fused eager aten ops on the
// packed path):
//   RMSNorm  transformers/models/llama/modeling_llama.py:62-67  (fp32 stats, cast, then * w)
//   residual transformers/models/llama/modeling_llama.py:306-324
//   SwiGLU   transformers/models/llama/modeling_llama.py:174-176
//   RoPE     transformers/models/llama/modeling_llama.py:113-160 with the packers'
//            restart-per-sentence position_ids (touchnet/models/llama/processing_llama.py:96-97)
//   LayerNorm/GELU: Whisper-style encoder layer driven by touchnet/models/qwen2_audio/__init__.py:18-133
//
// Layout: activations are row-major [rows, H] (rows = B*T of the packed buffer).  One wave64
// owns one row: every lane keeps its 16-byte vectors of the row in registers, statistics are
// wave-shuffle reductions (no LDS, no barrier), one HBM read + one write per element.
// Algorithmic bytes per row: see DESIGN.md §5.
#include "common.h"

namespace tn {

constexpr int kRowWaves = 4;  // rows per 256-thread block

// ------------------------------------------------------------------------------------------
// (residual-add +) RMSNorm forward.
//   h   = x (+ res_in)            rounded to T (HF adds in the activation dtype)
//   y   = T( w * T(h * rstd) )    rstd = rsqrt(mean(h^2) + eps) in fp32
// ------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res_in,
                                                          const T* __restrict__ w, T* __restrict__ y,
                                                          T* __restrict__ res_out, float* __restrict__ rstd_out,
                                                          int rows, int H, float eps) {
  constexpr int N = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = H / N;
  const size_t base = (size_t)row * H;
  float hv[MAXV][N];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 64 + lane;
    if (v < nvec) {
      Vec16<T> a;
      a.load(x + base + (size_t)v * N);
      a.unpack(hv[i]);
      if (res_in) {
        Vec16<T> r;
        float rf[N];
        r.load(res_in + base + (size_t)v * N);
        r.unpack(rf);
#pragma unroll
        for (int j = 0; j < N; ++j) hv[i][j] += rf[j];
        a.pack(hv[i]);  // round the sum to T ...
        a.unpack(hv[i]);  // ... and norm the rounded value, as the eager path does
        if (res_out) a.store(res_out + base + (size_t)v * N);
      }
#pragma unroll
      for (int j = 0; j < N; ++j) ss += hv[i][j] * hv[i][j];
    }
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss / (float)H + eps);
  if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 64 + lane;
    if (v < nvec) {
      Vec16<T> wv, o;
      float wf[N], of[N];
      wv.load(w + (size_t)v * N);
      wv.unpack(wf);
#pragma unroll
      for (int j = 0; j < N; ++j) of[j] = hv[i][j] * rstd;
      o.pack(of);  // T(h * rstd)
      o.unpack(of);
#pragma unroll
      for (int j = 0; j < N; ++j) of[j] *= wf[j];
      o.pack(of);
      o.store(y + base + (size_t)v * N);
    }
  }
}

// ------------------------------------------------------------------------------------------
// RMSNorm backward.  xhat = T(h*rstd);  g = dy*w;
//   dh = rstd * (g - xhat * mean(g*xhat)) (+ dres: gradient arriving on the residual stream)
//   dw_partial[block][c] = sum over the block's rows of dy * xhat      (deterministic 2-stage)
// ------------------------------------------------------------------------------------------
// block-wide sum of two values at once (one barrier pair); `sm` holds >= 2 * blockDim.x/64 floats
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sm) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) {
    sm[2 * w] = a;
    sm[2 * w + 1] = b;
  }
  __syncthreads();
  a = b = 0.f;
  for (int i = 0; i < nw; ++i) {
    a += sm[2 * i];
    b += sm[2 * i + 1];
  }
}

// One 256-thread block walks rows (grid-stride); thread t owns 16-byte vectors t, t+256, ... of every row,
// so its slice of dw accumulates in registers across rows and is written once per block.
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h,
                                                          const T* __restrict__ w, const float* __restrict__ rstd_in,
                                                          const T* __restrict__ dres, T* __restrict__ dh,
                                                          float* __restrict__ dw_partial, int rows, int H) {
  constexpr int N = Vec16<T>::N;
  __shared__ float sm[8];
  const int tid = threadIdx.x;
  const int nvec = H / N;
  float wf[MAXV][N], dwacc[MAXV][N];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 256 + tid;
#pragma unroll
    for (int j = 0; j < N; ++j) dwacc[i][j] = 0.f;
    if (v < nvec) {
      Vec16<T> wv;
      wv.load(w + (size_t)v * N);
      wv.unpack(wf[i]);
    }
  }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const size_t base = (size_t)row * H;
    const float rstd = rstd_in[row];
    float g[MAXV][N], xh[MAXV][N];
    float dot = 0.f, unused = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = i * 256 + tid;
      if (v < nvec) {
        Vec16<T> a, b;
        float dyf[N];
        a.load(dy + base + (size_t)v * N);
        a.unpack(dyf);
        b.load(h + base + (size_t)v * N);
        b.unpack(xh[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) xh[i][j] *= rstd;
        b.pack(xh[i]);
        b.unpack(xh[i]);  // xhat as the forward rounded it
#pragma unroll
        for (int j = 0; j < N; ++j) {
          dwacc[i][j] += dyf[j] * xh[i][j];
          g[i][j] = dyf[j] * wf[i][j];
          dot += g[i][j] * xh[i][j];
        }
      }
    }
    block_sum2(dot, unused, sm);
    dot /= (float)H;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = i * 256 + tid;
      if (v < nvec) {
        float o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = rstd * (g[i][j] - xh[i][j] * dot);
        if (dres) {
          Vec16<T> r;
          float rf[N];
          r.load(dres + base + (size_t)v * N);
          r.unpack(rf);
#pragma unroll
          for (int j = 0; j < N; ++j) o[j] += rf[j];
        }
        Vec16<T> ov;
        ov.pack(o);
        ov.store(dh + base + (size_t)v * N);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 256 + tid;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < N; ++j) dw_partial[(size_t)blockIdx.x * H + v * N + j] = dwacc[i][j];
    }
  }
}

// out[c] = sum_p partial[p][c]  (second stage of every weight/bias gradient).
// 16 columns per block (64-byte row segments), 16 row groups summed through LDS.  (The first version gave a block 64
// columns x 4 row groups: 64 workgroups for a 4096-wide gradient, 16 us per launch x 451 launches per step — most of
// the chip idle; with cols/16 workgroups the 1-4 MB of partials take a third of that.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ partial, T* __restrict__ out,
                                                              int P, int H) {
  __shared__ float sm[16][16 + 1];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < H) {
    int p = ry;
    for (; p + 48 < P; p += 64) {
      s0 += partial[(size_t)p * H + c];
      s1 += partial[(size_t)(p + 16) * H + c];
      s2 += partial[(size_t)(p + 32) * H + c];
      s3 += partial[(size_t)(p + 48) * H + c];
    }
    for (; p < P; p += 16) s0 += partial[(size_t)p * H + c];
  }
  sm[ry][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ry == 0 && c < H) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][cx];
    Elem<T>::st(out + c, t);
  }
}

// Bias gradient, stage 1:  partial[p][c] = sum over the rows of slab p of x[r][c]   (x bf16 [rows, ld], fp32 sums).
// Block = 256 columns (32 threads x 16-byte vectors) x 8 row lanes; slab p = rows [p*rps, (p+1)*rps).  torch's
// column reduction of a [30000, 1280] bf16 gradient runs at 1.8 TB/s; this pair of kernels is HBM-bound.
__global__ __launch_bounds__(256) void colsum_rows_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial,
                                                          int rows, int cols, long long ld, int rps) {
  __shared__ float sm[8][256 + 8];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 256 + cx * 8;
  const int r_beg = blockIdx.y * rps, r_end = min(r_beg + rps, rows);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {
    for (int r = r_beg + ry; r < r_end; r += 8) {
      Vec16<bf16_t> v;
      float f[8];
      v.load(x + (size_t)r * ld + c0);
      v.unpack(f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sm[ry][cx * 8 + j] = acc[j];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sm[k][threadIdx.x];
    partial[(size_t)blockIdx.y * cols + c] = t;
  }
}

// ------------------------------------------------------------------------------------------
// (residual-add +) LayerNorm forward:  y = T((h - mean) * rstd * w + b), fp32 statistics.
// ------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res_in,
                                                            const T* __restrict__ w, const T* __restrict__ b,
                                                            T* __restrict__ y, T* __restrict__ res_out,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int H, float eps) {
  constexpr int N = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * kRowWaves + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = H / N;
  const size_t base = (size_t)row * H;
  float hv[MAXV][N];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 64 + lane;
    if (v < nvec) {
      Vec16<T> a;
      a.load(x + base + (size_t)v * N);
      a.unpack(hv[i]);
      if (res_in) {
        Vec16<T> r;
        float rf[N];
        r.load(res_in + base + (size_t)v * N);
        r.unpack(rf);
#pragma unroll
        for (int j = 0; j < N; ++j) hv[i][j] += rf[j];
        a.pack(hv[i]);
        a.unpack(hv[i]);
        if (res_out) a.store(res_out + base + (size_t)v * N);
      }
#pragma unroll
      for (int j = 0; j < N; ++j) s += hv[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 64 + lane;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float d = hv[i][j] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 64 + lane;
    if (v < nvec) {
      Vec16<T> wv, bv, o;
      float wf[N], bf[N], of[N];
      wv.load(w + (size_t)v * N);
      wv.unpack(wf);
      bv.load(b + (size_t)v * N);
      bv.unpack(bf);
#pragma unroll
      for (int j = 0; j < N; ++j) of[j] = (hv[i][j] - mean) * rstd * wf[j] + bf[j];
      o.pack(of);
      o.store(y + base + (size_t)v * N);
    }
  }
}

// LayerNorm backward: xhat=(h-mean)*rstd; g=dy*w;
//   dh = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+dres);  dw=sum dy*xhat;  db=sum dy
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ h,
                                                            const T* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in,
                                                            const T* __restrict__ dres, T* __restrict__ dh,
                                                            float* __restrict__ dw_partial,
                                                            float* __restrict__ db_partial, int rows, int H) {
  constexpr int N = Vec16<T>::N;
  __shared__ float sm[8];
  const int tid = threadIdx.x;
  const int nvec = H / N;
  float wf[MAXV][N], dwacc[MAXV][N], dbacc[MAXV][N];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 256 + tid;
#pragma unroll
    for (int j = 0; j < N; ++j) dwacc[i][j] = dbacc[i][j] = 0.f;
    if (v < nvec) {
      Vec16<T> wv;
      wv.load(w + (size_t)v * N);
      wv.unpack(wf[i]);
    }
  }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const size_t base = (size_t)row * H;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float g[MAXV][N], xh[MAXV][N];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = i * 256 + tid;
      if (v < nvec) {
        Vec16<T> a, b;
        float dyf[N];
        a.load(dy + base + (size_t)v * N);
        a.unpack(dyf);
        b.load(h + base + (size_t)v * N);
        b.unpack(xh[i]);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          xh[i][j] = (xh[i][j] - mean) * rstd;
          dwacc[i][j] += dyf[j] * xh[i][j];
          dbacc[i][j] += dyf[j];
          g[i][j] = dyf[j] * wf[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
        }
      }
    }
    block_sum2(s1, s2, sm);
    s1 /= (float)H;
    s2 /= (float)H;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = i * 256 + tid;
      if (v < nvec) {
        float o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
        if (dres) {
          Vec16<T> r;
          float rf[N];
          r.load(dres + base + (size_t)v * N);
          r.unpack(rf);
#pragma unroll
          for (int j = 0; j < N; ++j) o[j] += rf[j];
        }
        Vec16<T> ov;
        ov.pack(o);
        ov.store(dh + base + (size_t)v * N);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = i * 256 + tid;
    if (v < nvec) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        dw_partial[(size_t)blockIdx.x * H + v * N + j] = dwacc[i][j];
        db_partial[(size_t)blockIdx.x * H + v * N + j] = dbacc[i][j];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Element-wise: SwiGLU and exact (erf) GELU, forward and backward, 16-byte vectors, grid-stride.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return sigmoid_fast(x); }   // (common.h: shared with the GEMM epilogues)

template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ gate, const T* __restrict__ up,
                                                         T* __restrict__ out, size_t nvec) {
  constexpr int N = Vec16<T>::N;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    Vec16<T> g, u, o;
    float gf[N], uf[N];
    g.load(gate + v * N);
    u.load(up + v * N);
    g.unpack(gf);
    u.unpack(uf);
#pragma unroll
    for (int j = 0; j < N; ++j) gf[j] = gf[j] * sigmoidf_(gf[j]);
    o.pack(gf);  // T(silu(g)) as the eager path materialises it
    o.unpack(gf);
#pragma unroll
    for (int j = 0; j < N; ++j) gf[j] *= uf[j];
    o.pack(gf);
    o.store(out + v * N);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ gate,
                                                         const T* __restrict__ up, T* __restrict__ dgate,
                                                         T* __restrict__ dup, size_t nvec) {
  constexpr int N = Vec16<T>::N;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    Vec16<T> d, g, u, og, ou;
    float df[N], gf[N], uf[N], dg[N], du[N];
    d.load(dout + v * N);
    g.load(gate + v * N);
    u.load(up + v * N);
    d.unpack(df);
    g.unpack(gf);
    u.unpack(uf);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float s = sigmoidf_(gf[j]);
      const float silu = gf[j] * s;
      du[j] = df[j] * silu;
      dg[j] = df[j] * uf[j] * (s + silu * (1.f - s));
    }
    og.pack(dg);
    ou.pack(du);
    og.store(dgate + v * N);
    ou.store(dup + v * N);
  }
}

// ---- SwiGLU with transposed second outputs (bf16): the MLP's weight-gradient GEMMs want act^T, d_gate^T and d_up^T
// (forward-layout GEMMs).  Writing them here costs one extra store of each tensor; the standalone
// transpose pass costs a load and a store.  A thread owns an 8 x 8 block (same tiling as transpose.hip): 16-byte row
// loads (8 adjacent lanes = one 128-byte line), v_perm register transposes, 16-byte stores both ways.
// Arithmetic identical to swiglu_fwd_kernel / swiglu_bwd_kernel (silu rounded to bf16 before the product).
__device__ __forceinline__ void transpose8x8_store(const uint4 (&in)[8], bf16_t* dst, long long dst_ld) {
  const uint32_t w[8][4] = {{in[0].x, in[0].y, in[0].z, in[0].w}, {in[1].x, in[1].y, in[1].z, in[1].w},
                            {in[2].x, in[2].y, in[2].z, in[2].w}, {in[3].x, in[3].y, in[3].z, in[3].w},
                            {in[4].x, in[4].y, in[4].z, in[4].w}, {in[5].x, in[5].y, in[5].z, in[5].w},
                            {in[6].x, in[6].y, in[6].z, in[6].w}, {in[7].x, in[7].y, in[7].z, in[7].w}};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int wi = c >> 1;
    const uint32_t sel = (c & 1) ? 0x07060302u : 0x05040100u;
    uint4 o;
    o.x = __builtin_amdgcn_perm(w[1][wi], w[0][wi], sel);
    o.y = __builtin_amdgcn_perm(w[3][wi], w[2][wi], sel);
    o.z = __builtin_amdgcn_perm(w[5][wi], w[4][wi], sel);
    o.w = __builtin_amdgcn_perm(w[7][wi], w[6][wi], sel);
    *reinterpret_cast<uint4*>(dst + (long long)c * dst_ld) = o;
  }
}

__global__ __launch_bounds__(256) void swiglu_fwd_t_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                                           bf16_t* __restrict__ out, bf16_t* __restrict__ out_t, int rows,
                                                           int cols) {
  const int cx = threadIdx.x & 7, rb = threadIdx.x >> 3;
  const int c0 = (blockIdx.x * 8 + cx) * 8, r0 = (blockIdx.y * 32 + rb) * 8;
  if (c0 >= cols || r0 >= rows) return;
  uint4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    Vec16<bf16_t> g, u, a;
    float gf[8], uf[8];
    g.load(gate + (size_t)(r0 + i) * cols + c0);
    u.load(up + (size_t)(r0 + i) * cols + c0);
    g.unpack(gf);
    u.unpack(uf);
#pragma unroll
    for (int j = 0; j < 8; ++j) gf[j] = gf[j] * sigmoidf_(gf[j]);
    a.pack(gf);
    a.unpack(gf);
#pragma unroll
    for (int j = 0; j < 8; ++j) gf[j] *= uf[j];
    a.pack(gf);
    a.store(out + (size_t)(r0 + i) * cols + c0);
    o[i] = a.raw;
  }
  transpose8x8_store(o, out_t + (size_t)c0 * rows + r0, rows);
}

__global__ __launch_bounds__(256) void swiglu_bwd_t_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ gate,
                                                           const bf16_t* __restrict__ up, bf16_t* __restrict__ dgate,
                                                           bf16_t* __restrict__ dup, bf16_t* __restrict__ dgu_t, int rows,
                                                           int cols) {
  const int cx = threadIdx.x & 7, rb = threadIdx.x >> 3;
  const int c0 = (blockIdx.x * 8 + cx) * 8, r0 = (blockIdx.y * 32 + rb) * 8;
  if (c0 >= cols || r0 >= rows) return;
  uint4 og[8], ou[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    Vec16<bf16_t> d, g, u, a, b;
    float df[8], gf[8], uf[8], dg[8], du[8];
    const size_t off = (size_t)(r0 + i) * cols + c0;
    d.load(dout + off);
    g.load(gate + off);
    u.load(up + off);
    d.unpack(df);
    g.unpack(gf);
    u.unpack(uf);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = sigmoidf_(gf[j]);
      const float silu = gf[j] * s;
      du[j] = df[j] * silu;
      dg[j] = df[j] * uf[j] * (s + silu * (1.f - s));
    }
    a.pack(dg);
    b.pack(du);
    a.store(dgate + off);
    b.store(dup + off);
    og[i] = a.raw;
    ou[i] = b.raw;
  }
  transpose8x8_store(og, dgu_t + (size_t)c0 * rows + r0, rows);                    // rows [0, cols): d_gate^T
  transpose8x8_store(ou, dgu_t + (size_t)(cols + c0) * rows + r0, rows);           // rows [cols, 2 cols): d_up^T
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, size_t nvec) {
  constexpr int N = Vec16<T>::N;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    Vec16<T> a;
    float f[N];
    a.load(x + v * N);
    a.unpack(f);
#pragma unroll
    for (int j = 0; j < N; j += 2) {
      const gelu_f32x2 r = gelu_f2(gelu_f32x2{f[j], f[j + 1]});
      f[j] = r.x;
      f[j + 1] = r.y;
    }
    a.pack(f);
    a.store(out + v * N);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                       T* __restrict__ dx, size_t nvec) {
  constexpr int N = Vec16<T>::N;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    Vec16<T> a, d;
    float f[N], df[N];
    a.load(x + v * N);
    d.load(dout + v * N);
    a.unpack(f);
    d.unpack(df);
#pragma unroll
    for (int j = 0; j < N; j += 2) {
      const gelu_f32x2 r = gelu_grad_f2(gelu_f32x2{f[j], f[j + 1]}, gelu_f32x2{df[j], df[j + 1]});
      df[j] = r.x;
      df[j + 1] = r.y;
    }
    d.pack(df);
    d.store(dx + v * N);
  }
}

// ------------------------------------------------------------------------------------------
// RoPE.  Table: cos/sin[n, D/2] = cos/sin(position_ids[n] * inv_freq[i]) computed in fp32 and
// stored in T (the eager path casts its fp32 table to the activation dtype).
// Apply (out of place or in place when y == x, half-split convention): for i < D/2
//   y[i]       = x[i]*c[i] - x[i+D/2]*s[i]
//   y[i+D/2]   = x[i+D/2]*c[i] + x[i]*s[i]
// `sign` = +1 forward, -1 backward (the transpose rotation).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void rope_table_kernel(const int64_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                  T* __restrict__ cos_t, T* __restrict__ sin_t, int n, int half, float scaling) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * half) return;
  const int i = idx % half;
  const float ang = (float)pos[idx / half] * inv_freq[i];
  float s, c;
  sincosf(ang, &s, &c);
  Elem<T>::st(cos_t + idx, c * scaling);
  Elem<T>::st(sin_t + idx, s * scaling);
}

// x: [n, heads, D] contiguous; one thread handles one 16-byte vector pair (lo half / hi half)
template <typename T>
__global__ __launch_bounds__(256) void rope_apply_kernel(const T* xq, const T* xk, T* yq, T* yk,
                                                         const T* __restrict__ cos_t, const T* __restrict__ sin_t,
                                                         int n, int hq, int hk, int D, float sign) {
  constexpr int N = Vec16<T>::N;
  const int half = D / 2, vph = half / N;  // vectors per half head
  const int heads = hq + hk;
  const size_t total = (size_t)n * heads * vph;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int v = idx % vph;
    const int hd = (idx / vph) % heads;
    const size_t tok = idx / ((size_t)vph * heads);
    const size_t off = (hd < hq) ? (tok * hq + hd) * D : (tok * hk + (hd - hq)) * D;
    const T* p = ((hd < hq) ? xq : xk) + off;
    T* o = ((hd < hq) ? yq : yk) + off;
    Vec16<T> lo, hi, cv, sv;
    float a[N], b[N], c[N], s[N], ya[N], yb[N];
    lo.load(p + v * N);
    hi.load(p + half + v * N);
    cv.load(cos_t + tok * half + v * N);
    sv.load(sin_t + tok * half + v * N);
    lo.unpack(a);
    hi.unpack(b);
    cv.unpack(c);
    sv.unpack(s);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      rope_rotate(a[j], b[j], c[j], sign * s[j], ya[j], yb[j]);
    }
    lo.pack(ya);
    hi.pack(yb);
    lo.store(o + v * N);
    hi.store(o + half + v * N);
  }
}

// scalar fallback for tiny head dims (D/2 not a multiple of the vector width)
template <typename T>
__global__ void rope_apply_scalar_kernel(const T* xq, const T* xk, T* yq, T* yk, const T* __restrict__ cos_t,
                                         const T* __restrict__ sin_t, int n, int hq, int hk, int D, float sign) {
  const int half = D / 2, heads = hq + hk;
  const size_t total = (size_t)n * heads * half;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = idx % half;
    const int hd = (idx / half) % heads;
    const size_t tok = idx / ((size_t)half * heads);
    const size_t off = (hd < hq) ? (tok * hq + hd) * D : (tok * hk + (hd - hq)) * D;
    const T* p = ((hd < hq) ? xq : xk) + off;
    T* o = ((hd < hq) ? yq : yk) + off;
    const float a = Elem<T>::ld(p + i), b = Elem<T>::ld(p + half + i);
    const float c = Elem<T>::ld(cos_t + tok * half + i), s = sign * Elem<T>::ld(sin_t + tok * half + i);
    float ya, yb;
    rope_rotate(a, b, c, s, ya, yb);
    Elem<T>::st(o + i, ya);
    Elem<T>::st(o + half + i, yb);
  }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
static inline int pick_maxv(int H, int N) {
  const int per_lane = (H / N + 63) / 64;
  if (per_lane <= 1) return 1;
  if (per_lane <= 2) return 2;
  if (per_lane <= 4) return 4;
  if (per_lane <= 8) return 8;
  if (per_lane <= 16) return 16;
  return -1;
}

#define TN_DISPATCH_MAXV(mv, ...)                      \
  switch (mv) {                                        \
    case 1: { constexpr int MAXV = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int MAXV = 2; __VA_ARGS__; } break;   \
    case 4: { constexpr int MAXV = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int MAXV = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int MAXV = 16; __VA_ARGS__; } break; \
    default: return TN_EINVAL;                         \
  }

template <typename T>
static int rmsnorm_fwd_t(const void* x, const void* res_in, const void* w, void* y, void* res_out, float* rstd,
                         int rows, int H, float eps, hipStream_t st) {
  constexpr int N = Vec16<T>::N;
  if (H % N || rows <= 0) return TN_EINVAL;
  const int mv = pick_maxv(H, N);
  dim3 grid((rows + kRowWaves - 1) / kRowWaves), block(256);
  TN_DISPATCH_MAXV(mv, hipLaunchKernelGGL((rmsnorm_fwd_kernel<T, MAXV>), grid, block, 0, st, (const T*)x,
                                          (const T*)res_in, (const T*)w, (T*)y, (T*)res_out, rstd, rows, H, eps));
  TN_LAUNCH_CHECK();
  return TN_OK;
}

static inline int norm_bwd_blocks(int rows) { return rows < 1024 ? rows : 1024; }

static inline int pick_maxv_block(int H, int N) {   // vectors per thread when 256 threads share a row
  const int per = (H / N + 255) / 256;
  return per <= 1 ? 1 : (per <= 2 ? 2 : (per <= 4 ? 4 : -1));
}

template <typename T>
static int rmsnorm_bwd_t(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dh,
                         void* dw, float* ws, int rows, int H, hipStream_t st) {
  constexpr int N = Vec16<T>::N;
  if (H % N || rows <= 0) return TN_EINVAL;
  const int mv = pick_maxv_block(H, N);
  const int nb = norm_bwd_blocks(rows);
  TN_DISPATCH_MAXV(mv, hipLaunchKernelGGL((rmsnorm_bwd_kernel<T, MAXV>), dim3(nb), dim3(256), 0, st, (const T*)dy,
                                          (const T*)h, (const T*)w, rstd, (const T*)dres, (T*)dh, ws, rows, H));
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL((colsum_partials_kernel<T>), dim3((H + 15) / 16), dim3(256), 0, st, ws, (T*)dw, nb, H);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

template <typename T>
static int layernorm_fwd_t(const void* x, const void* res_in, const void* w, const void* b, void* y, void* res_out,
                           float* mean, float* rstd, int rows, int H, float eps, hipStream_t st) {
  constexpr int N = Vec16<T>::N;
  if (H % N || rows <= 0) return TN_EINVAL;
  const int mv = pick_maxv(H, N);
  dim3 grid((rows + kRowWaves - 1) / kRowWaves), block(256);
  TN_DISPATCH_MAXV(mv, hipLaunchKernelGGL((layernorm_fwd_kernel<T, MAXV>), grid, block, 0, st, (const T*)x,
                                          (const T*)res_in, (const T*)w, (const T*)b, (T*)y, (T*)res_out, mean, rstd,
                                          rows, H, eps));
  TN_LAUNCH_CHECK();
  return TN_OK;
}

template <typename T>
static int layernorm_bwd_t(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                           const void* dres, void* dh, void* dw, void* db, float* ws, int rows, int H,
                           hipStream_t st) {
  constexpr int N = Vec16<T>::N;
  if (H % N || rows <= 0) return TN_EINVAL;
  const int mv = pick_maxv_block(H, N);
  const int nb = norm_bwd_blocks(rows);
  float* ws_b = ws + (size_t)nb * H;
  TN_DISPATCH_MAXV(mv, hipLaunchKernelGGL((layernorm_bwd_kernel<T, MAXV>), dim3(nb), dim3(256), 0, st, (const T*)dy,
                                          (const T*)h, (const T*)w, mean, rstd, (const T*)dres, (T*)dh, ws, ws_b,
                                          rows, H));
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL((colsum_partials_kernel<T>), dim3((H + 15) / 16), dim3(256), 0, st, ws, (T*)dw, nb, H);
  hipLaunchKernelGGL((colsum_partials_kernel<T>), dim3((H + 15) / 16), dim3(256), 0, st, ws_b, (T*)db, nb, H);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

static inline int ew_grid(size_t nvec) {
  size_t b = (nvec + 255) / 256;
  return (int)(b < 4096 ? (b ? b : 1) : 4096);
}

}  // namespace tn

using namespace tn;

#define TN_DTYPE_SWITCH(dtype, ...)                                  \
  if ((dtype) == 0) { typedef float T; __VA_ARGS__; }                \
  else if ((dtype) == 1) { typedef bf16_t T; __VA_ARGS__; }          \
  else return TN_EINVAL;

extern "C" {

int tn_norm_bwd_workspace_floats(int rows, int H) { return 2 * norm_bwd_blocks(rows) * H; }

int tn_rmsnorm_fwd(const void* x, const void* res_in, const void* w, void* y, void* res_out, float* rstd, int rows,
                   int H, float eps, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, return rmsnorm_fwd_t<T>(x, res_in, w, y, res_out, rstd, rows, H, eps, (hipStream_t)stream));
}

int tn_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dres, void* dh,
                   void* dw, float* workspace, int rows, int H, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, return rmsnorm_bwd_t<T>(dy, h, w, rstd, dres, dh, dw, workspace, rows, H,
                                                 (hipStream_t)stream));
}

int tn_layernorm_fwd(const void* x, const void* res_in, const void* w, const void* b, void* y, void* res_out,
                     float* mean, float* rstd, int rows, int H, float eps, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, return layernorm_fwd_t<T>(x, res_in, w, b, y, res_out, mean, rstd, rows, H, eps,
                                                   (hipStream_t)stream));
}

int tn_layernorm_bwd(const void* dy, const void* h, const void* w, const float* mean, const float* rstd,
                     const void* dres, void* dh, void* dw, void* db, float* workspace, int rows, int H, int dtype,
                     void* stream) {
  TN_DTYPE_SWITCH(dtype, return layernorm_bwd_t<T>(dy, h, w, mean, rstd, dres, dh, dw, db, workspace, rows, H,
                                                   (hipStream_t)stream));
}

int tn_swiglu_fwd(const void* gate, const void* up, void* out, long long n, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, {
    constexpr int N = Vec16<T>::N;
    if (n % N) return TN_EINVAL;
    const size_t nvec = (size_t)n / N;
    hipLaunchKernelGGL((swiglu_fwd_kernel<T>), dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)gate, (const T*)up, (T*)out, nvec);
    TN_LAUNCH_CHECK();
    return TN_OK;
  });
}

int tn_swiglu_bwd(const void* dout, const void* gate, const void* up, void* dgate, void* dup, long long n, int dtype,
                  void* stream) {
  TN_DTYPE_SWITCH(dtype, {
    constexpr int N = Vec16<T>::N;
    if (n % N) return TN_EINVAL;
    const size_t nvec = (size_t)n / N;
    hipLaunchKernelGGL((swiglu_bwd_kernel<T>), dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dout, (const T*)gate, (const T*)up, (T*)dgate, (T*)dup, nvec);
    TN_LAUNCH_CHECK();
    return TN_OK;
  });
}

// bf16 only; rows, cols multiples of 8; out_t [cols, rows]; dgu_t [2 cols, rows] = [d_gate^T ; d_up^T]
int tn_swiglu_fwd_t(const void* gate, const void* up, void* out, void* out_t, int rows, int cols, void* stream) {
  if (rows <= 0 || cols <= 0 || (rows | cols) & 7) return TN_EINVAL;
  dim3 grid((cols + 63) / 64, (rows + 255) / 256);
  hipLaunchKernelGGL(swiglu_fwd_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gate,
                     (const bf16_t*)up, (bf16_t*)out, (bf16_t*)out_t, rows, cols);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_swiglu_bwd_t(const void* dout, const void* gate, const void* up, void* dgate, void* dup, void* dgu_t, int rows,
                    int cols, void* stream) {
  if (rows <= 0 || cols <= 0 || (rows | cols) & 7) return TN_EINVAL;
  dim3 grid((cols + 63) / 64, (rows + 255) / 256);
  hipLaunchKernelGGL(swiglu_bwd_t_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)dgate, (bf16_t*)dup, (bf16_t*)dgu_t, rows, cols);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

int tn_gelu_fwd(const void* x, void* out, long long n, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, {
    constexpr int N = Vec16<T>::N;
    if (n % N) return TN_EINVAL;
    const size_t nvec = (size_t)n / N;
    hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                       (T*)out, nvec);
    TN_LAUNCH_CHECK();
    return TN_OK;
  });
}

int tn_gelu_bwd(const void* dout, const void* x, void* dx, long long n, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, {
    constexpr int N = Vec16<T>::N;
    if (n % N) return TN_EINVAL;
    const size_t nvec = (size_t)n / N;
    hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const T*)dout, (const T*)x, (T*)dx, nvec);
    TN_LAUNCH_CHECK();
    return TN_OK;
  });
}

int tn_rope_table(const long long* position_ids, const float* inv_freq, void* cos_t, void* sin_t, int n, int half,
                  float attention_scaling, int dtype, void* stream) {
  TN_DTYPE_SWITCH(dtype, {
    const size_t total = (size_t)n * half;
    hipLaunchKernelGGL((rope_table_kernel<T>), dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const int64_t*)position_ids, inv_freq, (T*)cos_t, (T*)sin_t, n, half, attention_scaling);
    TN_LAUNCH_CHECK();
    return TN_OK;
  });
}

int tn_rope_apply(const void* q, const void* k, void* q_out, void* k_out, const void* cos_t, const void* sin_t, int n,
                  int hq, int hk, int D, int backward, int dtype, void* stream) {
  if (D % 2) return TN_EINVAL;
  const float sign = backward ? -1.f : 1.f;
  TN_DTYPE_SWITCH(dtype, {
    constexpr int N = Vec16<T>::N;
    if ((D / 2) % N == 0) {
      const size_t total = (size_t)n * (hq + hk) * (D / 2 / N);
      hipLaunchKernelGGL((rope_apply_kernel<T>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)q, (const T*)k, (T*)q_out, (T*)k_out, (const T*)cos_t, (const T*)sin_t, n, hq, hk,
                         D, sign);
    } else {
      const size_t total = (size_t)n * (hq + hk) * (D / 2);
      hipLaunchKernelGGL((rope_apply_scalar_kernel<T>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                         (const T*)q, (const T*)k, (T*)q_out, (T*)k_out, (const T*)cos_t, (const T*)sin_t, n, hq, hk,
                         D, sign);
    }
    TN_LAUNCH_CHECK();
    return TN_OK;
  });
}

// out[c] = sum_r x[r][c]  for a bf16 [rows, cols] matrix with row stride ld (bias gradient of a linear layer):
// two deterministic stages through `ws` (tn_colsum_workspace_floats(rows, cols) floats).
// slabs: enough workgroups to keep HBM busy (~4096; the first version used 64 slabs = 320 workgroups on a
// [30000, 1280] gradient and ran at 1.8 TB/s, PMC-confirmed), at least 16 rows each
static int colsum_slabs(int rows, int cols) {
  const int cb = (cols + 255) / 256;
  int s = (4096 + cb - 1) / cb;
  const int smax = rows / 16 > 0 ? rows / 16 : 1;
  s = s > 1024 ? 1024 : s;
  s = s > smax ? smax : s;
  return s < 1 ? 1 : s;
}

long long tn_colsum_workspace_floats(int rows, int cols) { return (long long)colsum_slabs(rows, cols) * cols; }

int tn_colsum_bf16(const void* x, void* out, float* ws, int rows, int cols, long long ld, void* stream) {
  if (rows <= 0 || cols <= 0 || cols % 8 || ld % 8 || ld < cols || ((uintptr_t)x & 15)) return TN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int slabs = colsum_slabs(rows, cols);
  const int rps = (rows + slabs - 1) / slabs;
  hipLaunchKernelGGL(colsum_rows_kernel, dim3((cols + 255) / 256, slabs), dim3(256), 0, st, (const bf16_t*)x, ws, rows,
                     cols, ld, rps);
  TN_LAUNCH_CHECK();
  hipLaunchKernelGGL((colsum_partials_kernel<bf16_t>), dim3((cols + 15) / 16), dim3(256), 0, st, ws, (bf16_t*)out,
                     slabs, cols);
  TN_LAUNCH_CHECK();
  return TN_OK;
}

}  // extern "C"
