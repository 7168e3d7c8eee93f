// BIT-CD `BASE_Transformer` (/root/reference/models/bit_cd.py:802-934): the token path.
//
// Pixel side (two kernels per direction, everything else of a decoder layer's attention lives in 2 x 1024 floats per image):
//   semantic tokenizer (:857-865)   tokens[b][l][:] = sum_n softmax_n(x[b][n][:] . Wa[l]) x[b][n][:]
//   decoder cross-attention (:476-524 behind PreNorm2 / Residual2, :436-459) with the token side folded into two small matrices:
//       h = LayerNorm(x_n)                         q = Wq h          k_j = Wk LayerNorm(m_j)      v_j = Wv LayerNorm(m_j)
//       s[hd][j] = scale * q_hd . k_j,hd = scale * h . A[j][hd]              A[j][hd] = Wq_hd^T k_j,hd           (32 floats)
//       o = Wo concat_hd(sum_j p[hd][j] v_j,hd) = sum_{j,hd} p[hd][j] Bv[j][hd]   Bv[j][hd] = Wo_hd v_j,hd        (32 floats)
//   A pixel reads its 32 channels, does two 32x32 products against A / Bv held in LDS, a softmax over the token_len keys of each
//   head, and writes 32 channels: 128 B of HBM traffic per pixel in bf16 instead of the 4 KB of q and the attention output of the
//   reference's formulation (8 heads x 64 dims), and the head dimension (64 or 8) no longer appears on the pixel side at all.
// Token side (2 x token_len tokens per image pair): generic strided fp32 batched products + row softmax, composed by the plan
// (kurosiwo_amd/bitcd_plan.py).  fp32 throughout: a few hundred values per image steer every pixel.
#include "common.h"
#include "errors.h"
#include "../../include/ksmi.h"

namespace {

constexpr int TC = 32;     // token / pixel channels (dim of the transformer, bit_cd.py:829)
constexpr int TQ = 32;     // heads * token_len scores per pixel (8 x 4)

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <typename T>
__device__ __forceinline__ void load_row32(const T* p, float* f) {
  constexpr int V = ElemTraits<T>::kVec;
#pragma unroll
  for (int i = 0; i < TC / V; ++i) vec_unpack<T>(((const u32x4*)p)[i], f + i * V);
}
template <typename T>
__device__ __forceinline__ void store_row32(T* p, const float* f) {
  constexpr int V = ElemTraits<T>::kVec;
#pragma unroll
  for (int i = 0; i < TC / V; ++i) ((u32x4*)p)[i] = vec_pack<T>(f + i * V);
}

// ------------------------------------------------------------------------------------------------
// strided fp32 batched product: c[b1][b2][m][n] = alpha * sum_k a[b1][b2][m][k] b[b1][b2][k][n] + bias[n] (+ c)
// ------------------------------------------------------------------------------------------------
struct BmmArgs {
  const float *a, *b, *bias;
  float* c;
  int nb2, M, N, K;
  int64_t a_b1, a_b2, a_m, a_k, b_b1, b_b2, b_k, b_n, c_b1, c_b2, c_m, c_n;
  float alpha;
  int accumulate;
};
__global__ __launch_bounds__(256) void bmm_f32_kernel(const BmmArgs g, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i % g.N);
    int64_t r = i / g.N;
    const int m = (int)(r % g.M); r /= g.M;
    const int b2 = (int)(r % g.nb2);
    const int64_t b1 = r / g.nb2;
    const float* ap = g.a + b1 * g.a_b1 + b2 * g.a_b2 + m * g.a_m;
    const float* bp = g.b + b1 * g.b_b1 + b2 * g.b_b2 + n * g.b_n;
    // eight independent partial sums: the 16 loads of a step are issued together instead of one dependent load pair per multiply
    float p8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= g.K; k += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { av[j] = ap[(int64_t)(k + j) * g.a_k]; bv[j] = bp[(int64_t)(k + j) * g.b_k]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) p8[j] = fmaf(av[j], bv[j], p8[j]);
    }
    for (; k < g.K; ++k) p8[0] = fmaf(ap[(int64_t)k * g.a_k], bp[(int64_t)k * g.b_k], p8[0]);
    const float acc = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    float* cp = g.c + b1 * g.c_b1 + b2 * g.c_b2 + m * g.c_m + n * g.c_n;
    float v = g.alpha * acc + (g.bias ? g.bias[n] : 0.f);
    if (g.accumulate) v += *cp;
    *cp = v;
  }
}

// rows of n <= 64 values: y = softmax(scale * x); backward dx = scale * y * (dy - sum(y dy))
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, float* y, int64_t rows, int n, float scale) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const float* xp = x + r * n;
    float mx = -INFINITY;
    for (int j = 0; j < n; ++j) mx = fmaxf(mx, xp[j] * scale);
    float s = 0.f;
    for (int j = 0; j < n; ++j) s += expf(xp[j] * scale - mx);
    const float inv = 1.f / s;
    for (int j = 0; j < n; ++j) y[r * n + j] = expf(xp[j] * scale - mx) * inv;
  }
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* y, const float* dy, float* dx, int64_t rows, int n, float scale) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    float dot = 0.f;
    for (int j = 0; j < n; ++j) dot += y[r * n + j] * dy[r * n + j];
    for (int j = 0; j < n; ++j) dx[r * n + j] = scale * y[r * n + j] * (dy[r * n + j] - dot);
  }
}

// ------------------------------------------------------------------------------------------------
// semantic tokenizer, one workgroup per image: x [img][N][32] -> tokens[b][date*L + l][32] (+ pos), img = date*B + b
// stats[img][l] = {max, sum of exp} of the pixel softmax (the backward recomputes the weights from them)
// ------------------------------------------------------------------------------------------------
template <typename T, int L>
__global__ __launch_bounds__(256) void semantic_tokens_fwd_kernel(const T* x, const float* wa, const float* pos, float* tokens, float* stats,
                                                                  int B, int N) {
  __shared__ float red[4], was[L * TC], part[4][L * TC];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < L * TC; i += 256) was[i] = wa[i];
  __syncthreads();
  const T* xb = x + (int64_t)img * N * TC;
  float mx[L], sm[L];
#pragma unroll
  for (int l = 0; l < L; ++l) mx[l] = -INFINITY;
  for (int n = tid; n < N; n += 256) {
    float f[TC];
    load_row32<T>(xb + (int64_t)n * TC, f);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < TC; ++c) s = fmaf(f[c], was[l * TC + c], s);
      mx[l] = fmaxf(mx[l], s);
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) { mx[l] = block_max_256(mx[l], red); sm[l] = 0.f; }
  float acc[L][TC];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[l][c] = 0.f;
  for (int n = tid; n < N; n += 256) {
    float f[TC];
    load_row32<T>(xb + (int64_t)n * TC, f);
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < TC; ++c) s = fmaf(f[c], was[l * TC + c], s);
      const float e = expf(s - mx[l]);
      sm[l] += e;
#pragma unroll
      for (int c = 0; c < TC; ++c) acc[l][c] = fmaf(e, f[c], acc[l][c]);
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) sm[l] = block_sum_256(sm[l], red);
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const float v = wave_sum(acc[l][c]);
      if (lane == 0) part[wave][l * TC + c] = v;
    }
  __syncthreads();
  const int date = img / B, b = img - date * B;
  for (int i = tid; i < L * TC; i += 256) {
    const int l = i / TC;
    const float v = (part[0][i] + part[1][i] + part[2][i] + part[3][i]) / sm[l];
    const int row = date * L + l;
    tokens[((int64_t)b * 2 * L + row) * TC + (i - l * TC)] = v + (pos ? pos[row * TC + (i - l * TC)] : 0.f);
  }
  if (tid < L) { stats[((int64_t)img * L + tid) * 2] = mx[tid]; stats[((int64_t)img * L + tid) * 2 + 1] = sm[tid]; }
}

// backward: dtok [b][2L][32] (gradient of the encoder input), dx (+)= ..., dwa_part[img][L*32].  Four lanes share a pixel, one token
// each (lane & 3 = l): 32 weights + 32 token-gradient values + 32 accumulators per lane instead of L times that.
template <typename T, int L>
__global__ __launch_bounds__(256) void semantic_tokens_bwd_kernel(const T* x, const float* wa, const float* stats, const float* dtok, T* dx,
                                                                  float* dwa_part, int B, int N, int accumulate) {
  static_assert(L == 4, "lane mapping: four tokens");
  __shared__ float redl[4][L], part[4][L * TC];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = tid & 3, slot = tid >> 2;
  const int date = img / B, b = img - date * B;
  float wl[TC], dl_[TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) { wl[c] = wa[l * TC + c]; dl_[c] = dtok[((int64_t)b * 2 * L + date * L + l) * TC + c]; }
  const T* xb = x + (int64_t)img * N * TC;
  T* dxb = dx + (int64_t)img * N * TC;
  const float mx = stats[((int64_t)img * L + l) * 2], inv = 1.f / stats[((int64_t)img * L + l) * 2 + 1];
  // G[l] = sum_n p[n][l] g[n][l],  g[n][l] = dtok[l] . x[n]
  float G = 0.f;
  for (int n = slot; n < N; n += 64) {
    float f[TC];
    load_row32<T>(xb + (int64_t)n * TC, f);
    float s = 0.f, g = 0.f;
#pragma unroll
    for (int c = 0; c < TC; ++c) { s = fmaf(f[c], wl[c], s); g = fmaf(f[c], dl_[c], g); }
    G = fmaf(expf(s - mx) * inv, g, G);
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) G += __shfl_xor(G, o, 64);
  if (lane < L) redl[wave][lane] = G;
  __syncthreads();
  G = redl[0][l] + redl[1][l] + redl[2][l] + redl[3][l];
  float acc[TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) acc[c] = 0.f;
  for (int n = slot; n < N; n += 64) {
    float f[TC], d[TC];
    load_row32<T>(xb + (int64_t)n * TC, f);
    float s = 0.f, g = 0.f;
#pragma unroll
    for (int c = 0; c < TC; ++c) { s = fmaf(f[c], wl[c], s); g = fmaf(f[c], dl_[c], g); }
    const float p = expf(s - mx) * inv;
    const float dsc = p * (g - G);                       // gradient of the score x[n] . Wa[l]
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      float v = fmaf(p, dl_[c], dsc * wl[c]);
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      d[c] = v;
      acc[c] = fmaf(dsc, f[c], acc[c]);
    }
    if (l == 0) {
      if (accumulate) {
        float old[TC];
        load_row32<T>(dxb + (int64_t)n * TC, old);
#pragma unroll
        for (int c = 0; c < TC; ++c) d[c] += old[c];
      }
      store_row32<T>(dxb + (int64_t)n * TC, d);
    }
  }
#pragma unroll
  for (int c = 0; c < TC; ++c) {
    float v = acc[c];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if (lane < L) part[wave][lane * TC + c] = v;
  }
  __syncthreads();
  for (int i = tid; i < L * TC; i += 256) dwa_part[(int64_t)img * L * TC + i] = part[0][i] + part[1][i] + part[2][i] + part[3][i];
}

// ------------------------------------------------------------------------------------------------
// decoder cross-attention on the pixels of one image (blockIdx.y = img = date*B + b); token block of that image =
// rows (b*2 + date)*L .. +L of A / Bv ([row][heads][32], i.e. [L*heads][32] per image)
// ------------------------------------------------------------------------------------------------
template <typename T, int L>
__device__ __forceinline__ void cross_row_forward(const float* f, const float* gam, const float* bet, const float* As, float scale, float* h,
                                                  float* p, float* mean_out, float* rstd_out) {
  constexpr int HEADS = TQ / L;
  float mu = 0.f;
#pragma unroll
  for (int c = 0; c < TC; ++c) mu += f[c];
  mu *= 1.f / TC;
  float var = 0.f;
#pragma unroll
  for (int c = 0; c < TC; ++c) var = fmaf(f[c] - mu, f[c] - mu, var);
  const float rstd = rsqrtf(var * (1.f / TC) + 1e-5f);
#pragma unroll
  for (int c = 0; c < TC; ++c) h[c] = (f[c] - mu) * rstd * gam[c] + bet[c];
  *mean_out = mu; *rstd_out = rstd;
  // scores: index q = j*HEADS + hd
#pragma unroll
  for (int q = 0; q < TQ; ++q) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < TC; ++c) s = fmaf(h[c], As[q * TC + c], s);
    p[q] = s * scale;
  }
#pragma unroll
  for (int hd = 0; hd < HEADS; ++hd) {
    float mx = p[hd];
#pragma unroll
    for (int j = 1; j < L; ++j) mx = fmaxf(mx, p[j * HEADS + hd]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) { p[j * HEADS + hd] = expf(p[j * HEADS + hd] - mx); sum += p[j * HEADS + hd]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < L; ++j) p[j * HEADS + hd] *= inv;
  }
}

template <typename T, int L>
__global__ __launch_bounds__(256) void token_cross_fwd_kernel(const T* x, const float* gamma, const float* beta, const float* A, const float* Bv,
                                                              const float* bo, T* y, int B, int N, float scale) {
  // A / Bv in LDS, read as 64-lane broadcasts: the LDS return path (not the VALU) bounds this kernel, 25 us per launch at 2 x 32 x 3136
  // pixels; the same matrices through scalar loads (as the backward kernel does to stay under 256 VGPRs) measured 33 us
  __shared__ __attribute__((aligned(16))) float As[TQ * TC], Bs[TQ * TC], gam[TC], bet[TC], bos[TC];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int date = img / B, b = img - date * B;
  const int64_t tok = ((int64_t)b * 2 + date) * TQ * TC;
  for (int i = tid; i < TQ * TC; i += 256) { As[i] = A[tok + i]; Bs[i] = Bv[tok + i]; }
  if (tid < TC) { gam[tid] = gamma[tid]; bet[tid] = beta[tid]; bos[tid] = bo[tid]; }
  __syncthreads();
  const int n = blockIdx.x * 256 + tid;
  if (n >= N) return;
  const int64_t off = ((int64_t)img * N + n) * TC;
  float f[TC], h[TC], p[TQ], mu, rstd;
  load_row32<T>(x + off, f);
  cross_row_forward<T, L>(f, gam, bet, As, scale, h, p, &mu, &rstd);
#pragma unroll
  for (int c = 0; c < TC; ++c) h[c] = f[c] + bos[c];
#pragma unroll
  for (int q = 0; q < TQ; ++q)
#pragma unroll
    for (int c = 0; c < TC; ++c) h[c] = fmaf(p[q], Bs[q * TC + c], h[c]);
  store_row32<T>(y + off, h);
}

// backward of y = x + Wo attn(...) + bo for one block of 256 pixels: g (the gradient of the residual stream, [img][N][32]) is updated
// in place (g += LayerNorm^T A^T ds); part[blk][0..1023] = dA, [1024..2047] = dBv, [2048..2079] = dgamma, [..2111] = dbeta,
// [..2143] = dbo; blk = img * gridDim.x + blockIdx.x
constexpr int CROSS_PART = 2 * TQ * TC + 3 * TC;
template <typename T, int L>
__global__ __launch_bounds__(256) void token_cross_bwd_kernel(const T* x, const float* gamma, const float* beta, const float* A, const float* Bv,
                                                              T* g, float* part, int B, int N, float scale) {
  constexpr int HEADS = TQ / L;
  constexpr int LD = TC + 1;
  __shared__ float u[128 * LD], w[128 * LD];        // per pixel of a half block: (ds, h) then (p, do)
  const int img = blockIdx.y, tid = threadIdx.x;
  const int date = img / B, b = img - date * B;
  // uniform addresses -> scalar loads, the multiplies take SGPR operands: no LDS broadcast reads and 235 instead of 282 VGPRs
  // (two waves per SIMD): 163 -> 119 us per launch
  const float* __restrict__ As = A + ((int64_t)b * 2 + date) * TQ * TC;
  const float* __restrict__ Bs = Bv + ((int64_t)b * 2 + date) * TQ * TC;
  const float* __restrict__ gam = gamma;
  const float* __restrict__ bet = beta;
  const int n = blockIdx.x * 256 + tid;
  const bool live = n < N;
  const int64_t off = ((int64_t)img * N + (live ? n : 0)) * TC;
  float f[TC], h[TC], p[TQ], dy[TC], mu = 0.f, rstd = 0.f;
  if (live) {
    load_row32<T>(x + off, f);
    load_row32<T>(g + off, dy);
    cross_row_forward<T, L>(f, gam, bet, As, scale, h, p, &mu, &rstd);
  } else {
#pragma unroll
    for (int c = 0; c < TC; ++c) { f[c] = 0.f; h[c] = 0.f; dy[c] = 0.f; }
#pragma unroll
    for (int q = 0; q < TQ; ++q) p[q] = 0.f;
  }
  // dp[q] = do . Bv[q];  ds = scale * p * (dp - sum_j p dp) within each head
  float ds[TQ];
#pragma unroll
  for (int q = 0; q < TQ; ++q) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < TC; ++c) s = fmaf(dy[c], Bs[q * TC + c], s);
    ds[q] = s;
  }
#pragma unroll
  for (int hd = 0; hd < HEADS; ++hd) {
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < L; ++j) dot = fmaf(p[j * HEADS + hd], ds[j * HEADS + hd], dot);
#pragma unroll
    for (int j = 0; j < L; ++j) ds[j * HEADS + hd] = scale * p[j * HEADS + hd] * (ds[j * HEADS + hd] - dot);
  }
  // dh = A^T ds; LayerNorm backward; g += dx
  float dh[TC];
#pragma unroll
  for (int c = 0; c < TC; ++c) dh[c] = 0.f;
#pragma unroll
  for (int q = 0; q < TQ; ++q)
#pragma unroll
    for (int c = 0; c < TC; ++c) dh[c] = fmaf(ds[q], As[q * TC + c], dh[c]);
  float xh[TC], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < TC; ++c) {
    xh[c] = (f[c] - mu) * rstd;
    const float gg = dh[c] * gam[c];
    s1 += gg; s2 = fmaf(gg, xh[c], s2);
  }
  if (live) {
    float o[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) o[c] = dy[c] + rstd * (dh[c] * gam[c] - s1 * (1.f / TC) - xh[c] * s2 * (1.f / TC));
    store_row32<T>(g + off, o);
  }
  float* pb = part + ((int64_t)img * gridDim.x + blockIdx.x) * CROSS_PART;
  // dgamma, dbeta, dbo: block sums over the pixels (dead lanes hold zeros: dh = 0 there because ds = 0)
  {
    __shared__ float red3[4][3 * TC];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int c = 0; c < TC; ++c) {
      const float a = wave_sum(dh[c] * xh[c]), bsum = wave_sum(dh[c]), d = wave_sum(dy[c]);
      if (lane == 0) { red3[wave][c] = a; red3[wave][TC + c] = bsum; red3[wave][2 * TC + c] = d; }
    }
    __syncthreads();
    if (tid < 3 * TC) pb[2 * TQ * TC + tid] = red3[0][tid] + red3[1][tid] + red3[2][tid] + red3[3][tid];
  }
  // dA[q][c] = sum_pixels ds[q] h[c], dBv[q][c] = sum_pixels p[q] do[c]: stage the two factors of 128 pixels at a time, thread t owns
  // 4 consecutive outputs (q = t / 8, c = 4 (t % 8) ..)
  const int q = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if ((tid >> 7) == half) {
        const int r = tid & 127;
#pragma unroll
        for (int k = 0; k < TQ; ++k) u[r * LD + k] = which ? p[k] : ds[k];
#pragma unroll
        for (int c = 0; c < TC; ++c) w[r * LD + c] = which ? dy[c] : h[c];
      }
      __syncthreads();
      for (int r = 0; r < 128; ++r) {
        const float d = u[r * LD + q];
        a0 = fmaf(d, w[r * LD + c0], a0); a1 = fmaf(d, w[r * LD + c0 + 1], a1);
        a2 = fmaf(d, w[r * LD + c0 + 2], a2); a3 = fmaf(d, w[r * LD + c0 + 3], a3);
      }
    }
    float* o = pb + which * TQ * TC;
    o[q * TC + c0] = a0; o[q * TC + c0 + 1] = a1; o[q * TC + c0 + 2] = a2; o[q * TC + c0 + 3] = a3;
  }
}

// partial blocks -> dA / dBv per image ("=", token block (b*2+date)); the 96 LayerNorm / bias sums of an image go to small[img][96],
// token_cross_final_kernel adds the images up
__global__ __launch_bounds__(256) void token_cross_reduce_kernel(const float* part, float* dA, float* dBv, float* small, int B, int nblk) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= CROSS_PART) return;
  const int img = blockIdx.y, date = img / B, b = img - date * B;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += part[((int64_t)img * nblk + k) * CROSS_PART + i];
  const int64_t tok = ((int64_t)b * 2 + date) * TQ * TC;
  if (i < TQ * TC) dA[tok + i] = s;
  else if (i < 2 * TQ * TC) dBv[tok + i - TQ * TC] = s;
  else small[(int64_t)img * 3 * TC + i - 2 * TQ * TC] = s;
}
__global__ __launch_bounds__(128) void token_cross_final_kernel(const float* small, float* dgamma, float* dbeta, float* dbo, int images, int acc_ln,
                                                                int acc_bo) {
  const int i = threadIdx.x;
  if (i >= 3 * TC) return;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 4 <= images; k += 4)
#pragma unroll
    for (int j = 0; j < 4; ++j) s4[j] += small[(int64_t)(k + j) * 3 * TC + i];
  for (; k < images; ++k) s4[0] += small[(int64_t)k * 3 * TC + i];
  const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  if (i < TC) dgamma[i] = (acc_ln ? dgamma[i] : 0.f) + s;
  else if (i < 2 * TC) dbeta[i - TC] = (acc_ln ? dbeta[i - TC] : 0.f) + s;
  else dbo[i - 2 * TC] = (acc_bo ? dbo[i - 2 * TC] : 0.f) + s;
}

}  // namespace

#define KSMI_DT(dtype, EXPR_BF16, EXPR_F32)                                  \
  do {                                                                       \
    if ((dtype) == KSMI_BF16) { EXPR_BF16; }                                 \
    else if ((dtype) == KSMI_F32) { EXPR_F32; }                              \
    else return ksmi_fail(KSMI_E_ARG, "bad dtype");                          \
  } while (0)

extern "C" {

int ksmi_bmm_f32(const float* a, const float* b, const float* bias, float* c, int nb1, int nb2, int M, int N, int K, const int64_t* a_strides,
                 const int64_t* b_strides, const int64_t* c_strides, float alpha, int accumulate, void* stream) {
  if (!a || !b || !c || !a_strides || !b_strides || !c_strides || nb1 < 1 || nb2 < 1 || M < 1 || N < 1 || K < 1)
    return ksmi_fail(KSMI_E_ARG, "bmm_f32: bad argument");
  BmmArgs g{a, b, bias, c, nb2, M, N, K, a_strides[0], a_strides[1], a_strides[2], a_strides[3], b_strides[0], b_strides[1], b_strides[2],
            b_strides[3], c_strides[0], c_strides[1], c_strides[2], c_strides[3], alpha, accumulate};
  const int64_t total = (int64_t)nb1 * nb2 * M * N;
  const int64_t blocks = (total + 255) / 256;
  hipLaunchKernelGGL(bmm_f32_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream, g, total);
  return ksmi_check_launch("bmm_f32");
}

int ksmi_softmax_rows_f32(const float* x, float* y, int64_t rows, int n, float scale, void* stream) {
  if (rows < 1 || n < 1 || n > 64) return ksmi_fail(KSMI_E_ARG, "softmax_rows: 1 <= n <= 64");
  const int64_t blocks = (rows + 255) / 256;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream, x, y, rows, n, scale);
  return ksmi_check_launch("softmax_rows");
}

int ksmi_softmax_rows_backward_f32(const float* y, const float* dy, float* dx, int64_t rows, int n, float scale, void* stream) {
  if (rows < 1 || n < 1 || n > 64) return ksmi_fail(KSMI_E_ARG, "softmax_rows: 1 <= n <= 64");
  const int64_t blocks = (rows + 255) / 256;
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)(blocks > 65535 ? 65535 : blocks)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, rows, n, scale);
  return ksmi_check_launch("softmax_rows_backward");
}

static int token_shape_ok(int C, int L, int dtype) {
  if (C != TC) return ksmi_fail(KSMI_E_UNSUPPORTED, "BIT tokens: 32 channels (bit_cd.py:829)");
  if (L != 4) return ksmi_fail(KSMI_E_UNSUPPORTED, "BIT tokens: token_len 4 (every BASE_Transformer of define_G, bit_cd.py:690-700)");
  if (dtype != KSMI_F32 && dtype != KSMI_BF16) return ksmi_fail(KSMI_E_ARG, "bad dtype");
  return 0;
}

int ksmi_semantic_tokens_forward(const void* x, const float* wa, const float* pos, float* tokens, float* stats, int B, int dates, int N, int C,
                                 int L, int dtype, void* stream) {
  if (int rc = token_shape_ok(C, L, dtype)) return rc;
  if (dates != 2) return ksmi_fail(KSMI_E_ARG, "semantic_tokens: two dates");
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL((semantic_tokens_fwd_kernel<bf16_t, 4>), dim3(dates * B), dim3(256), 0, st, (const bf16_t*)x, wa, pos, tokens, stats, B, N),
          hipLaunchKernelGGL((semantic_tokens_fwd_kernel<float, 4>), dim3(dates * B), dim3(256), 0, st, (const float*)x, wa, pos, tokens, stats, B, N));
  return ksmi_check_launch("semantic_tokens_forward");
}

int ksmi_semantic_tokens_backward(const void* x, const float* wa, const float* stats, const float* dtokens, void* dx, float* dwa_partial, int B,
                                  int dates, int N, int C, int L, int accumulate, int dtype, void* stream) {
  if (int rc = token_shape_ok(C, L, dtype)) return rc;
  if (dates != 2) return ksmi_fail(KSMI_E_ARG, "semantic_tokens: two dates");
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL((semantic_tokens_bwd_kernel<bf16_t, 4>), dim3(dates * B), dim3(256), 0, st, (const bf16_t*)x, wa, stats, dtokens, (bf16_t*)dx, dwa_partial, B, N, accumulate),
          hipLaunchKernelGGL((semantic_tokens_bwd_kernel<float, 4>), dim3(dates * B), dim3(256), 0, st, (const float*)x, wa, stats, dtokens, (float*)dx, dwa_partial, B, N, accumulate));
  return ksmi_check_launch("semantic_tokens_backward");
}

int ksmi_token_cross_forward(const void* x, const float* gamma, const float* beta, const float* A, const float* Bv, const float* bo, void* y, int B,
                             int dates, int N, int C, int heads, int L, float scale, int dtype, void* stream) {
  if (int rc = token_shape_ok(C, L, dtype)) return rc;
  if (heads * L != TQ || dates != 2) return ksmi_fail(KSMI_E_UNSUPPORTED, "token_cross: 8 heads x 4 tokens, two dates");
  const dim3 grid((N + 255) / 256, dates * B);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL((token_cross_fwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, A, Bv, bo, (bf16_t*)y, B, N, scale),
          hipLaunchKernelGGL((token_cross_fwd_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, gamma, beta, A, Bv, bo, (float*)y, B, N, scale));
  return ksmi_check_launch("token_cross_forward");
}

size_t ksmi_token_cross_bwd_workspace(int B, int dates, int N) {
  return ((size_t)dates * B * ((N + 255) / 256) * CROSS_PART + (size_t)dates * B * 3 * TC) * sizeof(float);
}

int ksmi_token_cross_backward(const void* x, const float* gamma, const float* beta, const float* A, const float* Bv, void* g, float* dA, float* dBv,
                              float* dgamma, float* dbeta, float* dbo, int accumulate_ln, int accumulate_bo, float* workspace, int B, int dates,
                              int N, int C, int heads, int L, float scale, int dtype, void* stream) {
  if (int rc = token_shape_ok(C, L, dtype)) return rc;
  if (heads * L != TQ || dates != 2) return ksmi_fail(KSMI_E_UNSUPPORTED, "token_cross: 8 heads x 4 tokens, two dates");
  const int nblk = (N + 255) / 256;
  const dim3 grid(nblk, dates * B);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL((token_cross_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, A, Bv, (bf16_t*)g, workspace, B, N, scale),
          hipLaunchKernelGGL((token_cross_bwd_kernel<float, 4>), grid, dim3(256), 0, st, (const float*)x, gamma, beta, A, Bv, (float*)g, workspace, B, N, scale));
  int rc = ksmi_check_launch("token_cross_backward");
  if (rc) return rc;
  float* small = workspace + (size_t)dates * B * nblk * CROSS_PART;
  hipLaunchKernelGGL(token_cross_reduce_kernel, dim3((CROSS_PART + 255) / 256, dates * B), dim3(256), 0, st, workspace, dA, dBv, small, B, nblk);
  hipLaunchKernelGGL(token_cross_final_kernel, dim3(1), dim3(128), 0, st, small, dgamma, dbeta, dbo, dates * B, accumulate_ln, accumulate_bo);
  return ksmi_check_launch("token_cross_reduce");
}

}  // extern "C"
