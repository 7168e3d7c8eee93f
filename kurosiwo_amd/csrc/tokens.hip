// Token-sequence kernels of the FloodViT path (rows V1-V4 of SURVEY.md §8(a)); activations are
// [rows = B*N tokens][C] in T (bf16 / fp32), i.e. the same memory as NHWC with H*W = N.
// The token GEMMs (to_qkv, to_out, FFN) run on the implicit-GEMM kernels as 1x1 convolutions.
//
// Round-1 note: attention here is a straightforward VALU kernel (N = 197 keys live in LDS, one query row
// per lane, online softmax); it is ~5 % of the ViT FLOPs.  The MFMA flash-style version is a later step.
#include <cstdlib>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wave per row (vision_transformer.py:22,43,72,124,126; eps 1e-5)
// ------------------------------------------------------------------------------------------------
// sum over the LPR consecutive lanes that share a row (LPR = 8, 16, 32 or 64)
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LPR lanes per row, 64/LPR rows per wave (narrow rows: C = 64 keeps all lanes busy with 8 rows per wave), MAXV vectors per lane
template <typename T, int MAXV, int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* x, const float* gamma, const float* beta, T* y,
                                                            float* mean, float* rstd, int rows, int C, float eps) {
  constexpr int VEC = ElemTraits<T>::kVec, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane % LPR;
  const int row = (blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const bool rok = row < rows;
  const int nv = C / VEC;
  float v[MAXV][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int q = sl + i * LPR;
    if (rok && q < nv) {
      vec_unpack<T>(*(const u32x4*)(x + (size_t)row * C + q * VEC), v[i]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += v[i][j];
    }
  }
  const float mu = group_sum<LPR>(s) / (float)C;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (rok && sl + i * LPR < nv) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float dlt = v[i][j] - mu; q2 += dlt * dlt; }
    }
  const float rs = 1.0f / sqrtf(group_sum<LPR>(q2) / (float)C + eps);
  if (!rok) return;
  if (sl == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int q = sl + i * LPR;
    if (q < nv) {
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (v[i][j] - mu) * rs * gamma[q * VEC + j] + beta[q * VEC + j];
      *(u32x4*)(y + (size_t)row * C + q * VEC) = vec_pack<T>(o);
    }
  }
}

// dx = rstd*(g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); partial[blk][0][c] = sum dy, [1][c] = sum dy*xhat
template <typename T, int MAXV, int LPR>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const T* x, const float* mean, const float* rstd,
                                                            const float* gamma, T* dx, int accumulate, float* partial,
                                                            int rows, int C, int rows_per_block) {
  constexpr int VEC = ElemTraits<T>::kVec, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane % LPR, rslot = lane / LPR;
  const int nv = C / VEC;
  float gsum[MAXV][VEC], bsum[MAXV][VEC], gam[MAXV][VEC];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      gsum[i][j] = 0.f; bsum[i][j] = 0.f;
      const int q = sl + i * LPR;
      gam[i][j] = q < nv ? gamma[q * VEC + j] : 0.f;
    }
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  // the rows of a wave are walked with a one-row prefetch: the loads of row r + 4*RPW are in flight while row r is reduced (with
  // 13 rows per workgroup at ViT size the kernel was a chain of exposed load latencies: 17.6 us for 19 MB)
  u32x4 pdy[MAXV], px[MAXV];
  float pmu = 0.f, prs = 0.f;
  auto fetch = [&](int rb) {
    const int row = rb + rslot;
    const bool rok = row < r1;
    pmu = rok ? mean[row] : 0.f; prs = rok ? rstd[row] : 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int q = sl + i * LPR;
      const bool ok = rok && q < nv;
      pdy[i] = ok ? *(const u32x4*)(dy + (size_t)row * C + q * VEC) : (u32x4){0u, 0u, 0u, 0u};
      px[i] = ok ? *(const u32x4*)(x + (size_t)row * C + q * VEC) : (u32x4){0u, 0u, 0u, 0u};
    }
  };
  if (r0 + wave * RPW < r1) fetch(r0 + wave * RPW);
  for (int rb = r0 + wave * RPW; rb < r1; rb += 4 * RPW) {
    const int row = rb + rslot;
    const bool rok = row < r1;
    const float mu = pmu, rs = prs;
    u32x4 cdy[MAXV], cx[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { cdy[i] = pdy[i]; cx[i] = px[i]; }
    if (rb + 4 * RPW < r1) fetch(rb + 4 * RPW);
    float g[MAXV][VEC], xh[MAXV][VEC];
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int q = sl + i * LPR;
      if (rok && q < nv) {
        float xv[VEC];
        vec_unpack<T>(cdy[i], g[i]);
        vec_unpack<T>(cx[i], xv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          xh[i][j] = (xv[j] - mu) * rs;
          bsum[i][j] += g[i][j]; gsum[i][j] += g[i][j] * xh[i][j];
          const float gg = g[i][j] * gam[i][j];
          a += gg; b += gg * xh[i][j];
        }
      }
    }
    a = group_sum<LPR>(a) / (float)C; b = group_sum<LPR>(b) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int q = sl + i * LPR;
      if (rok && q < nv) {
        float o[VEC];
        T* dp = dx + (size_t)row * C + q * VEC;
        if (accumulate) vec_unpack<T>(*(const u32x4*)dp, o);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = rs * (g[i][j] * gam[i][j] - a - xh[i][j] * b);
          o[j] = accumulate ? o[j] + d : d;
        }
        *(u32x4*)dp = vec_pack<T>(o);
      }
    }
  }
  // combine the row slots of a wave (lanes sl, sl + LPR, ...) by shuffles, then the 4 waves through LDS: partial[blk][2][C]
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { bsum[i][j] += __shfl_xor(bsum[i][j], o, 64); gsum[i][j] += __shfl_xor(gsum[i][j], o, 64); }
  extern __shared__ float red[];                 // [2][C]
  for (int ph = 0; ph < 4; ++ph) {
    __syncthreads();
    if (wave == ph && rslot == 0) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int q = sl + i * LPR;
        if (q < nv)
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const int c = q * VEC + j;
            if (ph == 0) { red[c] = bsum[i][j]; red[C + c] = gsum[i][j]; }
            else { red[c] += bsum[i][j]; red[C + c] += gsum[i][j]; }
          }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += kThreads) partial[(size_t)blockIdx.x * 2 * C + i] = red[i];
}

// ------------------------------------------------------------------------------------------------
// GELU (exact erf form, nn.GELU default; vision_transformer.py:24), residual add
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
template <typename T, int MODE>   // 0: y = gelu(x) ; 1: dx = dy*gelu'(x) ; 2: out = a + b ; 3: relu bwd: dx = dy*(x>0) ; 4: y = relu(x)
__global__ void eltwise_kernel(const T* a, const T* b, T* out, int64_t nvec) {
  constexpr int VEC = ElemTraits<T>::kVec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    float x[VEC], y[VEC];
    vec_unpack<T>(*(const u32x4*)(a + v * VEC), x);
    if (MODE >= 1 && MODE <= 3) vec_unpack<T>(*(const u32x4*)(b + v * VEC), y);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (MODE == 0) x[j] = gelu_f(x[j]);
      else if (MODE == 4) x[j] = fmaxf(x[j], 0.f);
      else if (MODE == 1) x[j] = x[j] * gelu_df(y[j]);        // a = dy, b = x
      else if (MODE == 2) x[j] = x[j] + y[j];
      else x[j] = y[j] > 0.f ? x[j] : 0.f;                    // a = dy, b = x
    }
    *(u32x4*)(out + v * VEC) = vec_pack<T>(x);
  }
}

// dx = (dy * Dropout(site; element)) * gelu'(x): the backward of act -> drop (models/changeformer.py:129-130) in one pass; the masked
// gradient is rounded to the storage type before the product, as the two passes it replaces did (bit-identical)
template <typename T>
__global__ void gelu_bwd_drop_kernel(const T* dy, const T* x, T* dx, int64_t nvec, uint32_t thr, float inv, uint32_t site,
                                     const uint32_t* __restrict__ rng) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const uint32_t key = ksmi_rng_key(rng, site);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    float g[VEC], y[VEC];
    vec_unpack<T>(*(const u32x4*)(dy + v * VEC), g);
    vec_unpack<T>(*(const u32x4*)(x + v * VEC), y);
    const uint32_t e0 = (uint32_t)(v * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float m = ksmi_rng_keep(key, e0 + j, thr) ? ElemTraits<T>::cvt(g[j] * inv) : 0.f;
      g[j] = m * gelu_df(y[j]);
    }
    *(u32x4*)(dx + v * VEC) = vec_pack<T>(g);
  }
}

// ------------------------------------------------------------------------------------------------
// patchify: "b c (h p1) (w p2) -> b (h w) (p1 p2 c)"  (vision_transformer.py:122); image NCHW fp32
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void patchify_kernel(const float* x, T* out, int B, int Cin, int H, int W, int P) {
  const int gh = H / P, gw = W / P, D = P * P * Cin;
  const int64_t total = (int64_t)B * gh * gw * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = i % D; int64_t r = i / D;
    const int pw = r % gw; r /= gw;
    const int ph = r % gh; const int b = r / gh;
    const int c = k % Cin; const int pp = k / Cin;
    const int p2 = pp % P, p1 = pp / P;
    ElemTraits<T>::st(out + i, x[(((int64_t)b * Cin + c) * H + ph * P + p1) * W + pw * P + p2]);
  }
}

// x0[b,0,:] = cls + pos[0] ; x0[b,1+i,:] = emb[b,i,:] + pos[1+i]   (vision_transformer.py:143-145)
template <typename T>
__global__ void vit_embed_fwd_kernel(const T* emb, const float* cls, const float* pos, T* x0, int B, int N1, int C) {
  const int64_t total = (int64_t)B * N1 * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; int64_t r = i / C;
    const int n = r % N1; const int b = r / N1;
    const float v = n == 0 ? cls[c] : ElemTraits<T>::ld(emb + ((int64_t)b * (N1 - 1) + n - 1) * C + c);
    ElemTraits<T>::st(x0 + i, v + pos[(int64_t)n * C + c]);
  }
}
// dpos[n,c] = sum_b dx0[b,n,c] ; dcls[c] = sum_b dx0[b,0,c] ; demb[b,i,c] = dx0[b,1+i,c]
template <typename T>
__global__ void vit_embed_bwd_kernel(const T* dx0, T* demb, float* dcls, float* dpos, int B, int N1, int C) {
  const int64_t total = (int64_t)N1 * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; const int n = i / C;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float v = ElemTraits<T>::ld(dx0 + ((int64_t)b * N1 + n) * C + c);
      s += v;
      if (n > 0 && demb) ElemTraits<T>::st(demb + ((int64_t)b * (N1 - 1) + n - 1) * C + c, v);
    }
    dpos[i] = s;
    if (n == 0) dcls[c] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// multi-head attention, softmax(q k^T * scale) v   (vision_transformer.py:52-63); qkv [B*N][3*H*D]
// with the reference's "(3 h d)" channel order; out [B*N][H*D]; lse [B][H][N] (log-sum-exp of the scaled scores)
// ------------------------------------------------------------------------------------------------
// helpers on LDS rows of D elements (vector reads: every lane reads the SAME row -> LDS broadcast)
template <typename T, int D>
__device__ __forceinline__ float row_dot(const T* row, const float* a) {
  constexpr int VEC = ElemTraits<T>::kVec;
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < D / VEC; ++v) {
    float f[VEC];
    vec_unpack<T>(*(const u32x4*)(row + v * VEC), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s += a[v * VEC + j] * f[j];
  }
  return s;
}
template <typename T, int D>
__device__ __forceinline__ void row_axpy(float* acc, float w, const T* row) {
  constexpr int VEC = ElemTraits<T>::kVec;
#pragma unroll
  for (int v = 0; v < D / VEC; ++v) {
    float f[VEC];
    vec_unpack<T>(*(const u32x4*)(row + v * VEC), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[v * VEC + j] += w * f[j];
  }
}
template <typename T, int D>
__device__ __forceinline__ void row_load(float* dst, const T* row, float mul) {
  constexpr int VEC = ElemTraits<T>::kVec;
#pragma unroll
  for (int v = 0; v < D / VEC; ++v) {
    float f[VEC];
    vec_unpack<T>(*(const u32x4*)(row + v * VEC), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) dst[v * VEC + j] = f[j] * mul;
  }
}
template <typename T, int D>
__device__ __forceinline__ void row_store(T* row, const float* src, float mul) {
  constexpr int VEC = ElemTraits<T>::kVec;
#pragma unroll
  for (int v = 0; v < D / VEC; ++v) {
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] = src[v * VEC + j] * mul;
    *(u32x4*)(row + v * VEC) = vec_pack<T>(f);
  }
}
// cooperative copy of the [N][D] slice (head h, part 0/1/2 = q/k/v) of qkv into LDS
template <typename T, int D>
__device__ __forceinline__ void stage_rows(T* dst, const T* src, int N, int row_stride) {
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int VPR = D / VEC;
  for (int i = threadIdx.x; i < N * VPR; i += kThreads) {
    const int j = i / VPR, v = i - j * VPR;
    *(u32x4*)(dst + j * D + v * VEC) = *(const u32x4*)(src + (size_t)j * row_stride + v * VEC);
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* qkv, T* out, float* lse, int N, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* ks = (T*)smem;                    // [N][D]
  T* vs = ks + (size_t)N * D;          // [N][D]
  const int b = blockIdx.y, h = blockIdx.x;
  const int C3 = 3 * H * D;
  const T* base = qkv + (size_t)b * N * C3;
  stage_rows<T, D>(ks, base + (H + h) * D, N, C3);
  stage_rows<T, D>(vs, base + (2 * H + h) * D, N, C3);
  __syncthreads();
  for (int q = threadIdx.x; q < N; q += kThreads) {
    float qr[D], o[D];
    row_load<T, D>(qr, base + (size_t)q * C3 + h * D, scale);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) o[dd] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int j = 0; j < N; ++j) {
      const float s = row_dot<T, D>(ks + j * D, qr);
      const float mn = fmaxf(m, s);
      const float corr = expf(m - mn), p = expf(s - mn);
      l = l * corr + p;
#pragma unroll
      for (int dd = 0; dd < D; ++dd) o[dd] *= corr;
      row_axpy<T, D>(o, p, vs + j * D);
      m = mn;
    }
    row_store<T, D>(out + ((size_t)b * N + q) * (H * D) + h * D, o, 1.0f / l);
    lse[((size_t)b * H + h) * N + q] = m + logf(l);
  }
}

// backward, p recomputed from lse.  pass A (lane = query row): dq ; pass B1 (lane = key row): dv ; pass B2: dk.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* qkv, const T* out, const float* lse, const T* dout, T* dqkv,
                                                       int N, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // two LDS matrices at a time: pass A keeps K,V resident (q / dO rows come from global memory), passes B1/B2
  // keep Q,dO resident (k / v rows from global memory) -> 2*N*D elements instead of 4 (fp32 N=197 fits 160 KB)
  T* m0 = (T*)smem;                    // [N][D]  K, then Q
  T* m1 = m0 + (size_t)N * D;          // [N][D]  V, then dO
  float* dl = (float*)(m1 + (size_t)N * D);    // [N] D_q = sum_d dO*O
  float* ls = dl + N;                  // [N] lse
  const int b = blockIdx.y, h = blockIdx.x;
  const int C3 = 3 * H * D, C1 = H * D;
  const T* base = qkv + (size_t)b * N * C3;
  const T* dob = dout + (size_t)b * N * C1 + h * D;
  T* dbase = dqkv + (size_t)b * N * C3;
  stage_rows<T, D>(m0, base + (H + h) * D, N, C3);
  stage_rows<T, D>(m1, base + (2 * H + h) * D, N, C3);
  for (int q = threadIdx.x; q < N; q += kThreads) {
    float orow[D];
    row_load<T, D>(orow, out + ((size_t)b * N + q) * C1 + h * D, 1.f);
    dl[q] = row_dot<T, D>(dob + (size_t)q * C1, orow);
    ls[q] = lse[((size_t)b * H + h) * N + q];
  }
  __syncthreads();
  // pass A: dq[q] = scale * sum_j p*(dp - D_q) * k[j]
  for (int q = threadIdx.x; q < N; q += kThreads) {
    float qr[D], dor[D], dq[D];
    row_load<T, D>(qr, base + (size_t)q * C3 + h * D, scale);
    row_load<T, D>(dor, dob + (size_t)q * C1, 1.f);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) dq[dd] = 0.f;
    const float Lq = ls[q], Dq = dl[q];
    for (int j = 0; j < N; ++j) {
      const float s = row_dot<T, D>(m0 + j * D, qr);
      const float dp = row_dot<T, D>(m1 + j * D, dor);
      row_axpy<T, D>(dq, expf(s - Lq) * (dp - Dq), m0 + j * D);
    }
    row_store<T, D>(dbase + (size_t)q * C3 + h * D, dq, scale);
  }
  __syncthreads();
  stage_rows<T, D>(m0, base + h * D, N, C3);          // Q
  stage_rows<T, D>(m1, dob, N, C1);                    // dO
  __syncthreads();
  // pass B1: dv[j] = sum_q p[q][j] * dO[q]
  for (int j = threadIdx.x; j < N; j += kThreads) {
    float kr[D], dv[D];
    row_load<T, D>(kr, base + (size_t)j * C3 + (H + h) * D, scale);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) dv[dd] = 0.f;
    for (int q = 0; q < N; ++q) row_axpy<T, D>(dv, expf(row_dot<T, D>(m0 + q * D, kr) - ls[q]), m1 + q * D);
    row_store<T, D>(dbase + (size_t)j * C3 + (2 * H + h) * D, dv, 1.f);
  }
  // pass B2: dk[j] = scale * sum_q p*(dp - D_q) * q[q]
  for (int j = threadIdx.x; j < N; j += kThreads) {
    float kr[D], vr[D], dk[D];
    row_load<T, D>(kr, base + (size_t)j * C3 + (H + h) * D, scale);
    row_load<T, D>(vr, base + (size_t)j * C3 + (2 * H + h) * D, 1.f);
#pragma unroll
    for (int dd = 0; dd < D; ++dd) dk[dd] = 0.f;
    for (int q = 0; q < N; ++q) {
      const float p = expf(row_dot<T, D>(m0 + q * D, kr) - ls[q]);
      const float dp = row_dot<T, D>(m1 + q * D, vr);
      row_axpy<T, D>(dk, p * (dp - dl[q]), m0 + q * D);
    }
    row_store<T, D>(dbase + (size_t)j * C3 + (H + h) * D, dk, scale);
  }
}

// nearest x2 upsample with optional ReLU on the way (model_utilities.py:36-41: relu -> nn.Upsample(scale_factor=2))
template <typename T>
__global__ void upsample2_fwd_kernel(const T* x, T* y, int B, int H, int W, int C, int relu) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int64_t n = (int64_t)B * 2 * H * 2 * W * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ox = r % (2 * W); r /= 2 * W;
    const int oy = r % (2 * H); const int b = r / (2 * H);
    float f[VEC];
    vec_unpack<T>(*(const u32x4*)(x + (((int64_t)b * H + oy / 2) * W + ox / 2) * C + cv * VEC), f);
    if (relu) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = fmaxf(f[j], 0.f);
    }
    *(u32x4*)(y + v * VEC) = vec_pack<T>(f);
  }
}
// dx[b,y,x,c] = (x_pre > 0 if relu) * sum of the 4 dy positions
template <typename T>
__global__ void upsample2_bwd_kernel(const T* dy, const T* xpre, T* dx, int B, int H, int W, int C, int relu) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int64_t n = (int64_t)B * H * W * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ix = r % W; r /= W;
    const int iy = r % H; const int b = r / H;
    float s[VEC], t[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vec_unpack<T>(*(const u32x4*)(dy + (((int64_t)b * 2 * H + 2 * iy + (k >> 1)) * 2 * W + 2 * ix + (k & 1)) * C + cv * VEC), t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += t[j];
    }
    if (relu) {
      vec_unpack<T>(*(const u32x4*)(xpre + v * VEC), t);
#pragma unroll
      for (int j = 0; j < VEC; ++j) if (!(t[j] > 0.f)) s[j] = 0.f;
    }
    *(u32x4*)(dx + v * VEC) = vec_pack<T>(s);
  }
}

// ---- MAE token shuffles (models/mae.py:73-118) ------------------------------------------------------------------------
// gather : dst[b][j][:] = src[b][idx[b][j]][:] (+ table[idx[b][j] + table_off][:])           (:77-78, :82, :113)
// scatter: dst[b][idx[b][j]][:] = (src ? src[b][j][:] : fill[:]) + table[idx[b][j]][:]       (:94-110; the adjoint of gather
//          with table = fill = null).  idx rows are slices of a permutation, so no two sources hit one destination row.
template <typename T>
__global__ void gather_rows_kernel(const T* src, const int64_t* idx, int idx_rs, T* dst, const float* table, int table_off, int B, int Ns,
                                   int Nd, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int64_t n = (int64_t)B * Nd * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int j = r % Nd; const int b = r / Nd;
    const int t = (int)idx[(int64_t)b * idx_rs + j];
    float f[VEC];
    vec_unpack<T>(*(const u32x4*)(src + (((int64_t)b * Ns + t) * CV + cv) * VEC), f);
    if (table) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] += table[(int64_t)(t + table_off) * C + cv * VEC + e];
    }
    *(u32x4*)(dst + v * VEC) = vec_pack<T>(f);
  }
}

template <typename T>
__global__ void scatter_rows_kernel(const T* src, const float* fill, const int64_t* idx, int idx_rs, const float* table, T* dst, int B,
                                    int Nsrc, int Nd, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int64_t n = (int64_t)B * Nsrc * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int j = r % Nsrc; const int b = r / Nsrc;
    const int t = (int)idx[(int64_t)b * idx_rs + j];
    float f[VEC];
    if (src) vec_unpack<T>(*(const u32x4*)(src + v * VEC), f);
    else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = fill ? fill[cv * VEC + e] : 0.f;
    }
    if (table) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] += table[(int64_t)t * C + cv * VEC + e];
    }
    *(u32x4*)(dst + (((int64_t)b * Nd + t) * CV + cv) * VEC) = vec_pack<T>(f);
  }
}

// out[i] (+)= sum_b x[b][i], i < n (gradients of the position tables: every sample touches every row once); fixed order over b
template <typename T>
__global__ void batch_sum_kernel(const T* x, float* out, int B, int64_t n, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += ElemTraits<T>::ld(x + (int64_t)b * n + i);
    out[i] = accumulate ? out[i] + s : s;
  }
}

// F.mse_loss (models/mae.py:122): partial[block] = sum (pred - target)^2 over the block's elements, dpred = 2 (pred - target) scale
template <typename T>
__global__ void mse_kernel(const T* pred, const T* target, T* dpred, float* partial, int64_t nvec, float scale0, const float* upstream) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const float scale = scale0 * (upstream ? *upstream : 1.f);
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    float a[VEC], b[VEC], g[VEC];
    vec_unpack<T>(*(const u32x4*)(pred + v * VEC), a);
    vec_unpack<T>(*(const u32x4*)(target + v * VEC), b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { const float d = a[e] - b[e]; s += d * d; g[e] = 2.f * d * scale; }
    if (dpred) *(u32x4*)(dpred + v * VEC) = vec_pack<T>(g);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void mse_finish_kernel(const float* partial, int nblk, double count, float* loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)(red[0] / count);
}

// x[:, 1:] of the token sequence (vision_transformer.py:150-151) as a dense [B][N1-1][C] image, and its adjoint
template <typename T>
__global__ void drop_cls_kernel(const T* x, T* y, int B, int N1, int C, int backward) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int64_t n = (int64_t)B * N1 * CV;                      // indexed over the [B][N1][C] side
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int t = r % N1; const int b = r / N1;
    const int64_t full = v * VEC, dense = (((int64_t)b * (N1 - 1) + (t - 1)) * CV + cv) * VEC;
    if (!backward) { if (t > 0) *(u32x4*)(y + dense) = *(const u32x4*)(x + full); }
    else *(u32x4*)(y + full) = t > 0 ? *(const u32x4*)(x + dense) : (u32x4){0u, 0u, 0u, 0u};
  }
}

// logits NHWC [B][HW][Cs] (C real channels) <-> NCHW fp32 [B][C][HW]; the adjoint zero-fills the pad channels
template <typename T>
__global__ void logits_to_nchw_kernel(const T* x, float* y, int B, int C, int Cs, int64_t HW) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % HW; int64_t r = i / HW;
    const int c = r % C; const int b = r / C;
    y[i] = ElemTraits<T>::ld(x + ((int64_t)b * HW + p) * Cs + c);
  }
}
template <typename T>
__global__ void dlogits_to_nhwc_kernel(const float* dy, T* dx, int B, int C, int Cs, int64_t HW) {
  const int64_t n = (int64_t)B * HW * Cs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % Cs; int64_t r = i / Cs;
    const int64_t p = r % HW; const int b = r / HW;
    ElemTraits<T>::st(dx + i, c < C ? dy[((int64_t)b * C + c) * HW + p] : 0.f);
  }
}

template <typename T>
void attn_fwd_launch(dim3 grid, size_t lds, hipStream_t st, const void* qkv, void* out, float* lse, int N, int H, float scale) {
  auto k = attn_fwd_kernel<T, 64>;
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, (const T*)qkv, (T*)out, lse, N, H, scale);
}
template <typename T>
void attn_bwd_launch(dim3 grid, size_t lds, hipStream_t st, const void* qkv, const void* out, const float* lse, const void* dout,
                     void* dqkv, int N, int H, float scale) {
  auto k = attn_bwd_kernel<T, 64>;
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, (const T*)qkv, (const T*)out, lse, (const T*)dout, (T*)dqkv, N, H, scale);
}

int grid_for(int64_t n, int cap = 8192) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#define KSMI_DT(dtype, EXPR_BF16, EXPR_F32)                                  \
  do {                                                                       \
    if ((dtype) == KSMI_BF16) { EXPR_BF16; }                                 \
    else if ((dtype) == KSMI_F32) { EXPR_F32; }                              \
    else return ksmi_fail(KSMI_E_ARG, "bad dtype");                          \
  } while (0)

// (MAXV, LPR) by the number of 16-byte vectors per row
#define KSMI_LN_DISPATCH(nv_, CALL)                                                         \
  do {                                                                                      \
    if ((nv_) <= 8) { CALL(1, 8); }                                                         \
    else if ((nv_) <= 16) { CALL(1, 16); }                                                  \
    else if ((nv_) <= 32) { CALL(1, 32); }                                                  \
    else if ((nv_) <= 64) { CALL(1, 64); }                                                  \
    else if ((nv_) <= 128) { CALL(2, 64); }                                                 \
    else if ((nv_) <= 256) { CALL(4, 64); }                                                 \
    else { CALL(8, 64); }                                                                   \
  } while (0)

template <typename T>
void layernorm_fwd_launch(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int C,
                          float eps, hipStream_t st) {
  const int nv = C / ElemTraits<T>::kVec;
#define KSMI_LN_FWD(MAXV_, LPR_)                                                                                           \
  hipLaunchKernelGGL((layernorm_fwd_kernel<T, MAXV_, LPR_>), dim3((rows + 4 * (64 / LPR_) - 1) / (4 * (64 / LPR_))), dim3(256), 0, st, \
                     (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, eps)
  KSMI_LN_DISPATCH(nv, KSMI_LN_FWD);
#undef KSMI_LN_FWD
}

template <typename T>
void layernorm_bwd_launch(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx, int accumulate,
                          float* partial, int rows, int C, int nblk, int rpb, hipStream_t st) {
  const int nv = C / ElemTraits<T>::kVec;
  const size_t lds = (size_t)2 * C * sizeof(float);
#define KSMI_LN_BWD(MAXV_, LPR_)                                                                                           \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, MAXV_, LPR_>), dim3(nblk), dim3(256), lds, st, (const T*)dy, (const T*)x, mean, rstd, gamma, \
                     (T*)dx, accumulate, partial, rows, C, rpb)
  KSMI_LN_DISPATCH(nv, KSMI_LN_BWD);
#undef KSMI_LN_BWD
}

extern "C" {

int ksmi_layernorm_forward(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                           int rows, int C, float eps, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || C / vec > 64 * 8) return ksmi_fail(KSMI_E_ARG, "layernorm: C must be a multiple of the vector and <= 512 vectors");
  KSMI_DT(dtype, layernorm_fwd_launch<bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)stream),
          layernorm_fwd_launch<float>(x, gamma, beta, y, mean, rstd, rows, C, eps, (hipStream_t)stream));
  return ksmi_check_launch("layernorm_fwd");
}

// at most 256 partial rows: one workgroup per CU and the parameter-gradient reduction stays a single launch (no fold pass)
int ksmi_layernorm_bwd_blocks(int rows) {
  // 256 workgroups (one per CU) up to ViT-sized token matrices (measured faster there than 512 / 1024); the 200 k-row maps of the MiT
  // encoder's first stage need more waves in flight to stream (1024: ChangeFormer +0.5 %)
  static const int env = ksmi_knob_int("KSMI_LN_BWD_BLOCKS", 0);
  const int cap = env > 0 ? env : (rows >= 100000 ? 1024 : 256);
  int b = (rows + 7) / 8;
  return b > cap ? cap : (b < 1 ? 1 : b);
}

int ksmi_layernorm_backward(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                            int accumulate, float* partial, int rows, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || C / vec > 64 * 8) return ksmi_fail(KSMI_E_ARG, "layernorm_bwd: unsupported C");
  const int nblk = ksmi_layernorm_bwd_blocks(rows);
  const int rpb = (rows + nblk - 1) / nblk;
  KSMI_DT(dtype, layernorm_bwd_launch<bf16_t>(dy, x, mean, rstd, gamma, dx, accumulate, partial, rows, C, nblk, rpb, (hipStream_t)stream),
          layernorm_bwd_launch<float>(dy, x, mean, rstd, gamma, dx, accumulate, partial, rows, C, nblk, rpb, (hipStream_t)stream));
  return ksmi_check_launch("layernorm_bwd");
}

#define KSMI_ELT(MODE_, a_, b_, out_, n_, what_)                                                                       \
  const int vec = dtype == KSMI_BF16 ? 8 : 4;                                                                          \
  if ((n_) % vec) return ksmi_fail(KSMI_E_ARG, what_ ": element count must be a multiple of the 16-byte vector");      \
  const int64_t nvec = (n_) / vec;                                                                                     \
  KSMI_DT(dtype,                                                                                                        \
          hipLaunchKernelGGL((eltwise_kernel<bf16_t, MODE_>), dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)(a_), (const bf16_t*)(b_), (bf16_t*)(out_), nvec), \
          hipLaunchKernelGGL((eltwise_kernel<float, MODE_>), dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)(a_), (const float*)(b_), (float*)(out_), nvec)); \
  return ksmi_check_launch(what_);

int ksmi_gelu_forward(const void* x, void* y, int64_t n, int dtype, void* stream) { KSMI_ELT(0, x, x, y, n, "gelu_fwd") }
int ksmi_gelu_backward(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream) { KSMI_ELT(1, dy, x, dx, n, "gelu_bwd") }
int ksmi_gelu_backward_drop(const void* dy, const void* x, void* dx, int64_t n, uint32_t thr, float inv_keep, uint32_t site, const uint32_t* rng_state,
                            int dtype, void* stream) {
  if (!thr) return ksmi_gelu_backward(dy, x, dx, n, dtype, stream);
  if (!rng_state) return ksmi_fail(KSMI_E_ARG, "gelu_backward_drop: the rng state is required");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (n % vec || n >= ((int64_t)1 << 32)) return ksmi_fail(KSMI_E_ARG, "gelu_backward_drop: element count must be a multiple of the 16-byte vector and < 2^32");
  const int64_t nvec = n / vec;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(gelu_bwd_drop_kernel<bf16_t>, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, nvec, thr, inv_keep, site, rng_state),
          hipLaunchKernelGGL(gelu_bwd_drop_kernel<float>, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x, (float*)dx, nvec, thr, inv_keep, site, rng_state));
  return ksmi_check_launch("gelu_bwd_drop");
}
int ksmi_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream) { KSMI_ELT(2, a, b, out, n, "add") }
int ksmi_relu_backward(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream) { KSMI_ELT(3, dy, x, dx, n, "relu_bwd") }

int ksmi_relu_forward(const void* x, void* y, int64_t n, int dtype, void* stream) { KSMI_ELT(4, x, x, y, n, "relu_fwd") }

int ksmi_drop_cls(const void* x, void* y, int B, int N1, int C, int backward, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || N1 < 2) return ksmi_fail(KSMI_E_ARG, "drop_cls: C must be a multiple of the vector, N1 >= 2");
  const int64_t n = (int64_t)B * N1 * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(drop_cls_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, N1, C, backward),
          hipLaunchKernelGGL(drop_cls_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, N1, C, backward));
  return ksmi_check_launch("drop_cls");
}

int ksmi_gather_rows(const void* src, const int64_t* idx, int idx_rs, void* dst, const float* table, int table_off, int B, int Ns, int Nd, int C,
                     int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || B < 1 || Nd < 1 || Ns < 1) return ksmi_fail(KSMI_E_ARG, "gather_rows: C must be a multiple of the vector");
  const int64_t n = (int64_t)B * Nd * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, idx, idx_rs, (bf16_t*)dst, table, table_off, B, Ns, Nd, C),
          hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)src, idx, idx_rs, (float*)dst, table, table_off, B, Ns, Nd, C));
  return ksmi_check_launch("gather_rows");
}

int ksmi_scatter_rows(const void* src, const float* fill, const int64_t* idx, int idx_rs, const float* table, void* dst, int B, int Nsrc, int Nd,
                      int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || B < 1 || Nd < 1 || Nsrc < 1) return ksmi_fail(KSMI_E_ARG, "scatter_rows: C must be a multiple of the vector");
  const int64_t n = (int64_t)B * Nsrc * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(scatter_rows_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, fill, idx, idx_rs, table, (bf16_t*)dst, B, Nsrc, Nd, C),
          hipLaunchKernelGGL(scatter_rows_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)src, fill, idx, idx_rs, table, (float*)dst, B, Nsrc, Nd, C));
  return ksmi_check_launch("scatter_rows");
}

int ksmi_batch_sum(const void* x, float* out, int B, int64_t n, int accumulate, int dtype, void* stream) {
  if (B < 1 || n < 1) return ksmi_fail(KSMI_E_ARG, "batch_sum: bad args");
  KSMI_DT(dtype,
          hipLaunchKernelGGL(batch_sum_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, B, n, accumulate),
          hipLaunchKernelGGL(batch_sum_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, B, n, accumulate));
  return ksmi_check_launch("batch_sum");
}

size_t ksmi_mse_workspace(void) { return 1024 * sizeof(float); }

int ksmi_mse_loss(const void* pred, const void* target, void* dpred, float grad_scale, const float* upstream, float* loss, float* workspace,
                  int64_t n, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (n < 1 || n % vec || !loss || !workspace) return ksmi_fail(KSMI_E_ARG, "mse_loss: element count must be a multiple of the vector");
  const int64_t nvec = n / vec;
  int blocks = (int)((nvec + 255) / 256); if (blocks > 1024) blocks = 1024;
  const float scale = grad_scale / (float)n;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(mse_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, (const bf16_t*)target, (bf16_t*)dpred, workspace, nvec, scale, upstream),
          hipLaunchKernelGGL(mse_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)pred, (const float*)target, (float*)dpred, workspace, nvec, scale, upstream));
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, blocks, (double)n, loss);
  return ksmi_check_launch("mse_loss");
}

int ksmi_logits_to_nchw(const void* x, float* y, int B, int C, int Cs, int64_t HW, int dtype, void* stream) {
  const int64_t n = (int64_t)B * C * HW;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(logits_to_nchw_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, B, C, Cs, HW),
          hipLaunchKernelGGL(logits_to_nchw_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, y, B, C, Cs, HW));
  return ksmi_check_launch("logits_to_nchw");
}

int ksmi_dlogits_to_nhwc(const float* dy, void* dx, int B, int C, int Cs, int64_t HW, int dtype, void* stream) {
  const int64_t n = (int64_t)B * HW * Cs;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(dlogits_to_nhwc_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, (bf16_t*)dx, B, C, Cs, HW),
          hipLaunchKernelGGL(dlogits_to_nhwc_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, (float*)dx, B, C, Cs, HW));
  return ksmi_check_launch("dlogits_to_nhwc");
}

int ksmi_patchify(const float* x_nchw, void* out, int B, int Cin, int H, int W, int P, int dtype, void* stream) {
  if (H % P || W % P) return ksmi_fail(KSMI_E_ARG, "patchify: image not divisible by the patch size");
  const int64_t n = (int64_t)B * Cin * H * W;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x_nchw, (bf16_t*)out, B, Cin, H, W, P),
          hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x_nchw, (float*)out, B, Cin, H, W, P));
  return ksmi_check_launch("patchify");
}

int ksmi_vit_embed_forward(const void* emb, const float* cls, const float* pos, void* x0, int B, int N1, int C, int dtype, void* stream) {
  const int64_t n = (int64_t)B * N1 * C;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(vit_embed_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)emb, cls, pos, (bf16_t*)x0, B, N1, C),
          hipLaunchKernelGGL(vit_embed_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)emb, cls, pos, (float*)x0, B, N1, C));
  return ksmi_check_launch("vit_embed_fwd");
}

int ksmi_vit_embed_backward(const void* dx0, void* demb, float* dcls, float* dpos, int B, int N1, int C, int dtype, void* stream) {
  const int64_t n = (int64_t)N1 * C;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(vit_embed_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx0, (bf16_t*)demb, dcls, dpos, B, N1, C),
          hipLaunchKernelGGL(vit_embed_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)dx0, (float*)demb, dcls, dpos, B, N1, C));
  return ksmi_check_launch("vit_embed_bwd");
}

int ksmi_attention_forward(const void* qkv, void* out, float* lse, int B, int N, int H, int D, float scale, int dtype, void* stream) {
  if (D != 64) return ksmi_fail(KSMI_E_UNSUPPORTED, "attention: dim_head must be 64");
  static const bool valu = ksmi_knob_is_set("KSMI_ATTN_VALU");       // A/B switch
  if (dtype == KSMI_BF16 && N <= 208 && !valu) return ksmi_attn_mfma_vit(0, qkv, out, lse, nullptr, nullptr, nullptr, B, N, H, scale, stream);
  const size_t es = dtype == KSMI_BF16 ? 2 : 4;
  const size_t lds = 2 * (size_t)N * D * es;
  if (lds > 160 * 1024) return ksmi_fail(KSMI_E_UNSUPPORTED, "attention: sequence too long for the LDS-resident kernel");
  const dim3 grid(H, B);
  KSMI_DT(dtype, attn_fwd_launch<bf16_t>(grid, lds, (hipStream_t)stream, qkv, out, lse, N, H, scale),
          attn_fwd_launch<float>(grid, lds, (hipStream_t)stream, qkv, out, lse, N, H, scale));
  return ksmi_check_launch("attention_fwd");
}

size_t ksmi_attention_bwd_workspace(int B, int N, int H, int D, int dtype) {
  return dtype == KSMI_BF16 ? ksmi_attn_mfma_workspace(B, N, N, H, D) : 256;
}

int ksmi_attention_backward(const void* qkv, const void* out, const float* lse, const void* dout, void* dqkv, void* workspace, int B, int N,
                            int H, int D, float scale, int dtype, void* stream) {
  if (D != 64) return ksmi_fail(KSMI_E_UNSUPPORTED, "attention: dim_head must be 64");
  static const bool valu = ksmi_knob_is_set("KSMI_ATTN_VALU");
  if (dtype == KSMI_BF16 && N <= 208 && !valu) {
    if (!workspace) return ksmi_fail(KSMI_E_ARG, "attention_bwd: workspace required (ksmi_attention_bwd_workspace)");
    return ksmi_attn_mfma_vit(1, qkv, (void*)out, (float*)lse, dout, dqkv, workspace, B, N, H, scale, stream);
  }
  const size_t es = dtype == KSMI_BF16 ? 2 : 4;
  const size_t lds = 2 * (size_t)N * D * es + 2 * (size_t)N * sizeof(float);
  if (lds > 160 * 1024) return ksmi_fail(KSMI_E_UNSUPPORTED, "attention_bwd: sequence too long for the LDS-resident kernel");
  const dim3 grid(H, B);
  KSMI_DT(dtype, attn_bwd_launch<bf16_t>(grid, lds, (hipStream_t)stream, qkv, out, lse, dout, dqkv, N, H, scale),
          attn_bwd_launch<float>(grid, lds, (hipStream_t)stream, qkv, out, lse, dout, dqkv, N, H, scale));
  return ksmi_check_launch("attention_bwd");
}

int ksmi_upsample2_forward(const void* x, void* y, int B, int H, int W, int C, int relu, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "upsample2: C must be a multiple of the vector");
  const int64_t n = (int64_t)B * 4 * H * W * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(upsample2_fwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, relu),
          hipLaunchKernelGGL(upsample2_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, H, W, C, relu));
  return ksmi_check_launch("upsample2_fwd");
}

int ksmi_upsample2_backward(const void* dy, const void* x_pre, void* dx, int B, int H, int W, int C, int relu, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "upsample2: C must be a multiple of the vector");
  const int64_t n = (int64_t)B * H * W * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(upsample2_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x_pre, (bf16_t*)dx, B, H, W, C, relu),
          hipLaunchKernelGGL(upsample2_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)x_pre, (float*)dx, B, H, W, C, relu));
  return ksmi_check_launch("upsample2_bwd");
}

}  // extern "C"
