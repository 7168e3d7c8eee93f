// HBM-bound kernels of the SNUNet train step: BatchNorm glue of conv_block_nested,
// max-pool, first-layer conv, bias/channel reductions, optimiser, layout helpers.
// All activations NHWC in T (fp32 / bf16), accessed as 16-byte vectors; all
// statistics, parameters and gradients fp32 (reductions finish in fp64).
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

constexpr int kThreads = 256;

// A block of 256 threads walks pixels [p0, p1) of an NHWC tensor with CV = C/VEC
// channel-vectors per pixel.  Thread t owns channel-vector cv = t % CV and pixel lane
// pl = t / CV (NPL = 256 / CV lanes); threads beyond NPL*CV idle.
struct ChanWalk {
  int cv, pl, npl, active;
  __device__ ChanWalk(int CV) {
    npl = kThreads / CV; if (npl < 1) npl = 1;
    active = (int)threadIdx.x < npl * CV && CV <= kThreads;
    cv = threadIdx.x % CV; pl = threadIdx.x / CV;
  }
};

// Reduce per-thread accumulators val[K*VEC] (K quantities x VEC channels) over the pixel
// lanes of the block and write partial[(row*K + k)*C + c].
template <int K, int VEC>
__device__ __forceinline__ void block_channel_reduce(const float* val, const ChanWalk& w, int CV, int C, float* partial, int row) {
  __shared__ float red[kThreads * 2];
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int j0 = 0; j0 < VEC; j0 += 2) {
      __syncthreads();
      red[threadIdx.x * 2 + 0] = w.active ? val[k * VEC + j0] : 0.f;
      red[threadIdx.x * 2 + 1] = w.active ? val[k * VEC + j0 + 1] : 0.f;
      __syncthreads();
      if ((int)threadIdx.x < CV * 2) {
        const int cv = threadIdx.x >> 1, jj = threadIdx.x & 1;
        float s = 0.f;
        for (int pl = 0; pl < w.npl; ++pl) s += red[(pl * CV + cv) * 2 + jj];
        partial[((size_t)row * K + k) * C + cv * VEC + j0 + jj] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize
// ------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* partial, int rows, int Cpad, int C, double count,
                                   const float* gamma, const float* beta, float* rmean, float* rvar,
                                   int64_t* nbt, float momentum, float eps, int training,
                                   float* mean_o, float* rstd_o, float* scale_o, float* shift_o) {
  __shared__ double red[2][64][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  if (!training) {
    if (threadIdx.x < 16 && c < C) {
      const float rs = 1.0f / sqrtf(rvar[c] + eps);
      const float sc = gamma[c] * rs;
      mean_o[c] = rmean[c]; rstd_o[c] = rs; scale_o[c] = sc; shift_o[c] = beta[c] - rmean[c] * sc;
    }
    return;
  }
  double s = 0.0, q = 0.0;
  if (c < C)
#pragma unroll 8
    for (int r = rl; r < rows; r += 64) {
      s += (double)partial[((size_t)r * 2 + 0) * Cpad + c];
      q += (double)partial[((size_t)r * 2 + 1) * Cpad + c];
    }
  red[0][rl][cl] = s; red[1][rl][cl] = q;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    s = 0.0; q = 0.0;
    for (int r = 0; r < 64; ++r) { s += red[0][r][cl]; q += red[1][r][cl]; }
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rs;
    mean_o[c] = (float)mean; rstd_o[c] = rs; scale_o[c] = sc; shift_o[c] = beta[c] - (float)mean * sc;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
}

// Stage 1 of long row reductions, IN PLACE: block (channel group, r) folds rows r, r+R, r+2R, ... into row r
// (fp64 accumulate), so the latency-bound finalize kernels below only walk R rows instead of thousands.
__global__ void rows_fold_kernel(float* partial, int rows, int K, int Cstride, int C, int R) {
  __shared__ double red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl, r0 = blockIdx.y;
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    if (c < C)
      for (int r = r0 + R * rl; r < rows; r += R * 16) s += (double)partial[((size_t)r * K + k) * Cstride + c];
    __syncthreads();
    red[rl][cl] = s;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
      s = 0.0;
      for (int r = 0; r < 16; ++r) s += red[r][cl];
      partial[((size_t)r0 * K + k) * Cstride + c] = (float)s;
    }
  }
}

static int fold_rows(float* partial, int rows, int K, int Cstride, int C, hipStream_t st) {
  constexpr int R = 32;
  static const int fold_min = ksmi_knob_int("KSMI_FOLD_MIN", 8) * R;
  if (rows <= fold_min) return rows;              // up to 256 rows: the finishing kernel's 64 row lanes walk 4 rows each, no extra launch
  hipLaunchKernelGGL(rows_fold_kernel, dim3((C + 15) / 16, R), dim3(256), 0, st, partial, rows, K, Cstride, C, R);
  return R;
}

// sums[k][c] = sum_rows partial[row][k][c]; optional accumulate into dgamma (k=1) / dbeta (k=0)
__global__ void reduce_rows_kernel(const float* partial, int rows, int K, int Cstride, int C, float* sums,
                                   float* dgamma, float* dbeta, int accumulate, float alpha) {
  __shared__ double red[64][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    if (c < C)
#pragma unroll 8
      for (int r = rl; r < rows; r += 64) s += (double)partial[((size_t)r * K + k) * Cstride + c];
    __syncthreads();
    red[rl][cl] = s;
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
      s = 0.0;
      for (int r = 0; r < 64; ++r) s += red[r][cl];
      const float f = (float)s;
      if (sums) sums[k * C + c] = f;
      float* tgt = (k == 0) ? dbeta : (k == 1 ? dgamma : nullptr);
      if (tgt) tgt[c] = accumulate ? tgt[c] + alpha * f : alpha * f;   // sums stay unscaled (the apply kernel takes alpha itself)
    }
  }
}

// all deferred row reductions of a step: block (entry, 16-channel group); 16 row lanes x 16 channel lanes, fp64, fixed order
__global__ __launch_bounds__(256) void reduce_rows_batched_kernel(const ksmi_rowsum_desc* descs) {
  __shared__ double red[16][17];
  const ksmi_rowsum_desc d0 = descs[blockIdx.x];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.y * 16 + cl;
  if (!d0.head || blockIdx.y * 16 >= d0.C) return;
  double s = 0.0;
  int e = blockIdx.x;
  while (e >= 0) {                                   // the chain of entries writing this destination, in order
    const ksmi_rowsum_desc d = descs[e];
    if (c < d.C) {
      double sa = 0.0, sb = 0.0;
      int r = rl;
      for (; r + 16 < d.rows; r += 32) {
        sa += (double)d.partial[((size_t)r * d.K + d.k) * d.Cstride + c];
        sb += (double)d.partial[((size_t)(r + 16) * d.K + d.k) * d.Cstride + c];
      }
      if (r < d.rows) sa += (double)d.partial[((size_t)r * d.K + d.k) * d.Cstride + c];
      s += sa + sb;
    }
    e = d.next;
  }
  red[rl][cl] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < d0.C) {
    s = 0.0;
    for (int r = 0; r < 16; ++r) s += red[r][cl];
    d0.dst[c] = d0.accumulate ? d0.dst[c] + (float)s : (float)s;
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise glue
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_add_relu_kernel(const T* z, const T* idn, const float* scale, const float* shift, T* out,
                                   int64_t nvec, int CV) {
  constexpr int VEC = ElemTraits<T>::kVec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CV) * VEC;
    float a[VEC], b[VEC];
    vec_unpack<T>(*(const u32x4*)(z + v * VEC), a);
    vec_unpack<T>(*(const u32x4*)(idn + v * VEC), b);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j] * scale[c + j] + shift[c + j] + b[j], 0.f);
    *(u32x4*)(out + v * VEC) = vec_pack<T>(a);
  }
}

// pass 1 of block backward: partial[row][0] = sum g, partial[row][1] = sum g*zhat, g = dout*(out>0)
template <typename T>
__global__ void bnrelu_bwd_reduce_kernel(const T* dout, const T* out, const T* z, const float* mean, const float* rstd,
                                         float* partial, int64_t npix, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  ChanWalk w(CV);
  float acc[2 * VEC];
#pragma unroll
  for (int j = 0; j < 2 * VEC; ++j) acc[j] = 0.f;
  if (w.active) {
    float mu[VEC], rs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = mean[w.cv * VEC + j]; rs[j] = rstd[w.cv * VEC + j]; }
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = per * blockIdx.x, p1 = min(npix, p0 + per);
    // two pixels per trip: six 16-byte loads in flight per lane (the grid is only 2 workgroups per CU)
    for (int64_t p = p0 + w.pl; p < p1; p += 2 * w.npl) {
      const bool two = p + w.npl < p1;
      const int64_t off0 = p * C + w.cv * VEC, off1 = two ? off0 + (int64_t)w.npl * C : off0;
      const u32x4 rg0 = *(const u32x4*)(dout + off0), ro0 = *(const u32x4*)(out + off0), rz0 = *(const u32x4*)(z + off0);
      const u32x4 rg1 = *(const u32x4*)(dout + off1), ro1 = *(const u32x4*)(out + off1), rz1 = *(const u32x4*)(z + off1);
      float g[VEC], o[VEC], zz[VEC];
      vec_unpack<T>(rg0, g); vec_unpack<T>(ro0, o); vec_unpack<T>(rz0, zz);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gg = o[j] > 0.f ? g[j] : 0.f;
        acc[j] += gg; acc[VEC + j] += gg * (zz[j] - mu[j]) * rs[j];
      }
      vec_unpack<T>(rg1, g); vec_unpack<T>(ro1, o); vec_unpack<T>(rz1, zz);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gg = (two && o[j] > 0.f) ? g[j] : 0.f;
        acc[j] += gg; acc[VEC + j] += gg * (zz[j] - mu[j]) * rs[j];
      }
    }
  }
  block_channel_reduce<2, VEC>(acc, w, CV, C, partial, blockIdx.x);
}

// pass 2: g -> dout (in place), dz = gamma*rstd*(g - s0/n - zhat*s1/n)
template <typename T>
__global__ void bnrelu_bwd_apply_kernel(T* dout, const T* out, const T* z, const float* mean, const float* rstd,
                                        const float* gamma, const float* sums, T* dz, float inv_n, int64_t nvec, int CV, int C,
                                        float alpha) {
  constexpr int VEC = ElemTraits<T>::kVec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CV) * VEC;
    float g[VEC], o[VEC], zz[VEC];
    vec_unpack<T>(*(const u32x4*)(dout + v * VEC), g);
    vec_unpack<T>(*(const u32x4*)(out + v * VEC), o);
    vec_unpack<T>(*(const u32x4*)(z + v * VEC), zz);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float gg = o[j] > 0.f ? g[j] : 0.f;
      const float zh = (zz[j] - mean[c + j]) * rstd[c + j];
      g[j] = gg;
      zz[j] = alpha * gamma[c + j] * rstd[c + j] * (gg - sums[c + j] * inv_n - zh * sums[C + c + j] * inv_n);
    }
    *(u32x4*)(dout + v * VEC) = vec_pack<T>(g);
    *(u32x4*)(dz + v * VEC) = vec_pack<T>(zz);
  }
}

// di = g + gamma*rstd*(r - t0/n - xhat*t1/n) written over r ; partial[row][0][c] = sum di
template <typename T>
__global__ void bn_bwd_apply_add_kernel(T* r, const T* g, const T* iv, const float* mean, const float* rstd,
                                        const float* gamma, const float* sums, float* partial, float inv_n,
                                        int64_t npix, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  ChanWalk w(CV);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (w.active) {
    float mu[VEC], rs[VEC], k0[VEC], k1[VEC], gr[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = w.cv * VEC + j;
      mu[j] = mean[c]; rs[j] = rstd[c]; gr[j] = gamma[c] * rstd[c];
      k0[j] = sums[c] * inv_n; k1[j] = sums[C + c] * inv_n;
    }
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = per * blockIdx.x, p1 = min(npix, p0 + per);
    for (int64_t p = p0 + w.pl; p < p1; p += 2 * w.npl) {   // two pixels per trip, all six loads issued before the first store
      const bool two = p + w.npl < p1;
      const int64_t offs[2] = {p * C + w.cv * VEC, two ? (p + w.npl) * C + w.cv * VEC : p * C + w.cv * VEC};
      u32x4 raw[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        raw[u][0] = *(const u32x4*)(r + offs[u]); raw[u][1] = *(const u32x4*)(g + offs[u]); raw[u][2] = *(const u32x4*)(iv + offs[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (u == 1 && !two) break;
        float rr[VEC], gg[VEC], ii[VEC];
        vec_unpack<T>(raw[u][0], rr); vec_unpack<T>(raw[u][1], gg); vec_unpack<T>(raw[u][2], ii);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xh = (ii[j] - mu[j]) * rs[j];
          const float di = gg[j] + gr[j] * (rr[j] - k0[j] - xh * k1[j]);
          rr[j] = di;
        }
        const u32x4 pk = vec_pack<T>(rr);
        *(u32x4*)(r + offs[u]) = pk;
        vec_unpack<T>(pk, rr);                 // bias gradient sums what the next kernels will read
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += rr[j];
      }
    }
  }
  block_channel_reduce<1, VEC>(acc, w, CV, C, partial, blockIdx.x);
}

// pass 2 when the producer of d out already gated it (conv gate epilogue, ksmi.h): dz = gamma*rstd*(g - s0/n - zhat*s1/n); g is only read
template <typename T>
__global__ void bn_bwd_apply_gated_kernel(const T* g, const T* z, const float* mean, const float* rstd, const float* gamma,
                                          const float* sums, T* dz, float inv_n, int64_t nvec, int CV, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CV) * VEC;
    float gg[VEC], zz[VEC];
    vec_unpack<T>(*(const u32x4*)(g + v * VEC), gg);
    vec_unpack<T>(*(const u32x4*)(z + v * VEC), zz);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float zh = (zz[j] - mean[c + j]) * rstd[c + j];
      zz[j] = gamma[c + j] * rstd[c + j] * (gg[j] - sums[c + j] * inv_n - zh * sums[C + c + j] * inv_n);
    }
    *(u32x4*)(dz + v * VEC) = vec_pack<T>(zz);
  }
}

template <typename T>
__global__ void channel_sum_kernel(const T* x, float* partial, int64_t npix, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  ChanWalk w(CV);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (w.active) {
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = per * blockIdx.x, p1 = min(npix, p0 + per);
    int64_t p = p0 + w.pl;
    for (; p + 3 * w.npl < p1; p += 4 * w.npl) {                  // four loads in flight, summed in the order of the plain loop
      u32x4 r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = *(const u32x4*)(x + (p + k * w.npl) * C + w.cv * VEC);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xx[VEC];
        vec_unpack<T>(r[k], xx);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += xx[j];
      }
    }
    for (; p < p1; p += w.npl) {
      float xx[VEC];
      vec_unpack<T>(*(const u32x4*)(x + p * C + w.cv * VEC), xx);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += xx[j];
    }
  }
  block_channel_reduce<1, VEC>(acc, w, CV, C, partial, blockIdx.x);
}

// wide rows (token matrices, C up to several thousand): block (row slab, 64-vector channel group); thread = (pixel lane
// t>>6, channel vector t&63) -> 1 KB contiguous per pixel row
template <typename T>
__global__ void channel_sum_wide_kernel(const T* x, float* partial, int64_t npix, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  __shared__ float red[4][64][VEC + 1];
  const int CV = C / VEC;
  const int cv = blockIdx.y * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (cv < CV) {
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = per * blockIdx.x, p1 = min(npix, p0 + per);
    for (int64_t p = p0 + pl; p < p1; p += 4) {
      float xx[VEC];
      vec_unpack<T>(*(const u32x4*)(x + p * C + cv * VEC), xx);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += xx[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[pl][threadIdx.x & 63][j] = acc[j];
  __syncthreads();
  if (pl == 0 && cv < CV) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      partial[(size_t)blockIdx.x * C + cv * VEC + j] = red[0][threadIdx.x][j] + red[1][threadIdx.x][j] + red[2][threadIdx.x][j] + red[3][threadIdx.x][j];
  }
}

// ------------------------------------------------------------------------------------------------
// max pool 2x2 stride 2
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(const T* x, T* y, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
  const int64_t n = (int64_t)B * Ho * Wo * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    const T* base = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + cv * VEC;
    float m[VEC], t[VEC];
    vec_unpack<T>(*(const u32x4*)base, m);
    vec_unpack<T>(*(const u32x4*)(base + C), t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = t[j] > m[j] ? t[j] : m[j];
    vec_unpack<T>(*(const u32x4*)(base + (int64_t)W * C), t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = t[j] > m[j] ? t[j] : m[j];
    vec_unpack<T>(*(const u32x4*)(base + (int64_t)W * C + C), t);
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = t[j] > m[j] ? t[j] : m[j];
    *(u32x4*)(y + v * VEC) = vec_pack<T>(m);
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* x, const T* dy, T* dx, int accumulate, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC, Ho = H / 2, Wo = W / 2;
  const int64_t n = (int64_t)B * Ho * Wo * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    const int64_t o00 = (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + cv * VEC;
    const int64_t offs[4] = {o00, o00 + C, o00 + (int64_t)W * C, o00 + (int64_t)W * C + C};
    float xv[4][VEC], g[VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k) vec_unpack<T>(*(const u32x4*)(x + offs[k]), xv[k]);
    vec_unpack<T>(*(const u32x4*)(dy + v * VEC), g);
    int am[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float m = xv[0][j]; am[j] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) if (xv[k][j] > m) { m = xv[k][j]; am[j] = k; }   // first maximum wins
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o[VEC];
      if (accumulate) vec_unpack<T>(*(const u32x4*)(dx + offs[k]), o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float gg = am[j] == k ? g[j] : 0.f;
        o[j] = accumulate ? o[j] + gg : gg;
      }
      *(u32x4*)(dx + offs[k]) = vec_pack<T>(o);
    }
  }
}

// The Dataset's per-tile pipeline folded into the image load of the first layer (dataset/Dataset.py:164-168 clamp to
// [0, clamp_input] + nan_to_num(clamp_input), :193-198 Normalize): nmean == nullptr -> the image is already normalised.  nclamp[c] < 0
// marks a channel without clamp (DEM / slope: NaN -> the mean, i.e. 0 after normalisation).  Same fp32 operations, same order as
// sar_preprocess_kernel (cformer.hip): the fused path is bit-identical to preprocess-then-convolve.
__device__ __forceinline__ float raw_tile_value(float v, int c, const float* nmean, const float* nstd, const float* nclamp) {
  if (nmean == nullptr) return v;
  const float hi = nclamp[c];
  if (hi >= 0.f) v = (v != v) ? hi : fminf(fmaxf(v, 0.f), hi);
  else if (v != v) v = nmean[c];
  return (v - nmean[c]) / nstd[c];
}

// Channel concat of the trainer's input assembly (change_detection_trainer.py:117-133: torch.cat((image, dem), dim=1)) as an address
// choice: channels [0, chead) come from x [B,chead,H,W], the rest from xtail [B,Cin-chead,H,W] (xtail == nullptr: chead = Cin).
__device__ __forceinline__ float image_value(const float* __restrict__ x, const float* __restrict__ xtail, int chead, int Cin, int b,
                                             int c, int H, int W, int iy, int ix) {
  return c < chead ? x[(((int64_t)b * chead + c) * H + iy) * W + ix] : xtail[(((int64_t)b * (Cin - chead) + (c - chead)) * H + iy) * W + ix];
}

// ------------------------------------------------------------------------------------------------
// first-layer conv (raw image NCHW fp32, Cin <= 8) -> NHWC T, + BN partial stats
// one 16x16 output patch per block, one pixel per thread
// ------------------------------------------------------------------------------------------------
template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* out, float* stats, int B,
                                                             int H, int W, int Cout, const float* __restrict__ nmean,
                                                             const float* __restrict__ nstd, const float* __restrict__ nclamp,
                                                             const float* __restrict__ xtail, int chead) {
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int KT = CIN * 9;
  constexpr int NC = 16, OP = NC + 1;             // channels per pass through the LDS tile (22 KB per workgroup: 7 per CU), its row pitch
  constexpr int G = 256 / NC;                     // pixel groups of the column sums, NC pixels each
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xs = (float*)smem;                       // [CIN][18][18]
  float* ot = xs + CIN * 324;                     // [256 pixels][OP]: the pass's outputs, fp32
  float* red = ot + 256 * OP;                     // [G pixel groups][2][NC]
  constexpr int KTP = (KT + 3) / 4 * 4;
  float* wl = red + G * 2 * NC;                   // [NC][KTP] weights of the pass + [NC] bias: broadcast 16-byte LDS reads
  const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
  int bm = blockIdx.x;
  const int tx = bm % tilesX; bm /= tilesX;
  const int ty = bm % tilesY; const int b = bm / tilesY;
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  for (int i = tid; i < CIN * 324; i += 256) {
    const int c = i / 324, r = i - c * 324, hy = r / 18, hx = r - hy * 18;
    const int iy = ty * 16 - 1 + hy, ix = tx * 16 - 1 + hx;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = raw_tile_value(image_value(x, xtail, chead, CIN, b, c, H, W, iy, ix), c, nmean, nstd, nclamp);
    xs[i] = v;
  }
  __syncthreads();
  const int oy = ty * 16 + ly, ox = tx * 16 + lx;
  const bool ok = oy < H && ox < W;
  float xin[KT];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) xin[c * 9 + t] = xs[c * 324 + (ly + t / 3) * 18 + lx + t % 3];
  // One pixel per thread computes NC channels into the LDS tile; the tile then leaves as whole 16-byte vectors of consecutive
  // pixels (a thread's own 32 channels are 4 scattered stores) and its column sums give the BatchNorm partial row of the block
  // in a fixed order (the per-channel wave shuffles of the first version were 2 x 6 LDS round trips per channel: 130 us per
  // 224 x 224 x 32 batch-32 image against 25 us of multiply-adds).
  for (int n0 = 0; n0 < Cout; n0 += NC) {
    const int nc = min(NC, Cout - n0);
    for (int i = tid; i < nc * KTP; i += 256) { const int j = i / KTP, k = i - j * KTP; wl[i] = k < KT ? w[(n0 + j) * KT + k] : 0.f; }
    if (tid < nc) wl[NC * KTP + tid] = bias ? bias[n0 + tid] : 0.f;
    __syncthreads();                               // (also orders this pass's tile writes behind the previous pass's readers)
#pragma unroll 4
    for (int j = 0; j < nc; ++j) {
      float a = wl[NC * KTP + j];
      float wr[KTP];
#pragma unroll
      for (int k = 0; k < KTP; k += 4) *(f32x4*)(wr + k) = *(const f32x4*)(wl + j * KTP + k);
#pragma unroll
      for (int k = 0; k < KT; ++k) a = fmaf(xin[k], wr[k], a);     // (one rounding per tap, in tap order: the first version's v_fmac chain)
      ot[tid * OP + j] = ok ? ElemTraits<T>::cvt(a) : 0.f;         // the STORED value (bf16 mode: rounded here, so the statistics below describe it)
    }
    __syncthreads();
    if (stats) {
      const int n = tid & (NC - 1), grp = tid / NC;
      if (n < nc) {
        float sm = 0.f, sq = 0.f;
#pragma unroll 8
        for (int i = 0; i < NC; ++i) { const float v = ot[(grp * NC + i) * OP + n]; sm += v; sq += v * v; }
        red[(grp * 2 + 0) * NC + n] = sm; red[(grp * 2 + 1) * NC + n] = sq;
      }
    }
    const int vpp = nc / VEC;                      // vectors per pixel in this pass
    for (int v = tid; v < 256 * vpp; v += 256) {
      const int p = v / vpp, q = v - p * vpp;
      const int py = ty * 16 + (p >> 4), px = tx * 16 + (p & 15);
      if (py >= H || px >= W) continue;
      float o[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = ot[p * OP + q * VEC + j];
      *(u32x4*)(out + (((int64_t)b * H + py) * W + px) * Cout + n0 + q * VEC) = vec_pack<T>(o);
    }
    __syncthreads();
    if (stats && tid < 2 * nc) {
      const int which = tid / nc, n = tid - which * nc;
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) t += red[(g * 2 + which) * NC + n];
      stats[((size_t)blockIdx.x * 2 + which) * Cout + n0 + n] = t;
    }
  }
}

// im2col of the raw image for the first 3x3 conv: out[b,y,x, c*9 + t] = x[b,c,y+t/3-1,x+t%3-1] (zero padded,
// channels >= Cin*9 zero).  The first conv and its weight gradient then run on the MFMA implicit-GEMM kernels
// as a 1x1 convolution over these Kpad "channels" (k = c*9 + t matches the OIHW flattening of the weight).
template <typename T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, T* out, int B, int Cin, int H, int W, int Kpad,
                                                        const float* __restrict__ nmean, const float* __restrict__ nstd,
                                                        const float* __restrict__ nclamp, const float* __restrict__ xtail, int chead) {
  constexpr int VEC = ElemTraits<T>::kVec;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xs = (float*)smem;                       // [Cin][18][18]
  const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
  int bm = blockIdx.x;
  const int tx = bm % tilesX; bm /= tilesX;
  const int ty = bm % tilesY; const int b = bm / tilesY;
  const int tid = threadIdx.x;
  for (int i = tid; i < Cin * 324; i += 256) {
    const int c = i / 324, r = i - c * 324, hy = r / 18, hx = r - hy * 18;
    const int iy = ty * 16 - 1 + hy, ix = tx * 16 - 1 + hx;
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = raw_tile_value(image_value(x, xtail, chead, Cin, b, c, H, W, iy, ix), c, nmean, nstd, nclamp);
    xs[i] = v;
  }
  __syncthreads();
  const int KV = Kpad / VEC;                      // vectors per pixel
  for (int v = tid; v < 256 * KV; v += 256) {
    const int p = v / KV, q = v - p * KV;
    const int ly = p >> 4, lx = p & 15;
    const int oy = ty * 16 + ly, ox = tx * 16 + lx;
    if (oy >= H || ox >= W) continue;
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int k = q * VEC + j;
      const int c = k / 9, t = k - c * 9;
      f[j] = c < Cin ? xs[c * 324 + (ly + t / 3) * 18 + lx + t % 3] : 0.f;
    }
    *(u32x4*)(out + (((int64_t)b * H + oy) * W + ox) * Kpad + q * VEC) = vec_pack<T>(f);
  }
}

// first-layer weight gradient: partial[blk][n][c*9+t] = sum over the block's patches
template <typename T>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* x, const T* dy, float* partial,
                                                               int B, int Cin, int H, int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* xs = (float*)smem;                       // [Cin][18][18]
  float* dys = xs + Cin * 324;                    // [256][Cout+1]
  const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
  const int patches = B * tilesX * tilesY;
  const int tid = threadIdx.x;
  const int KT = Cin * 9;
  const int n = tid % Cout, grp = tid / Cout, ngrp = 256 / Cout;   // Cout in {8,16,32,64,128,256}
  constexpr int MAXK = 12;                        // ceil(72 / ngrp) for Cout<=32 ; guarded below
  float acc[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; ++j) acc[j] = 0.f;
  const int ldy = Cout + 1;
  for (int patch = blockIdx.x; patch < patches; patch += gridDim.x) {
    int bm = patch;
    const int tx = bm % tilesX; bm /= tilesX;
    const int ty = bm % tilesY; const int b = bm / tilesY;
    __syncthreads();
    for (int i = tid; i < Cin * 324; i += 256) {
      const int c = i / 324, r = i - c * 324, hy = r / 18, hx = r - hy * 18;
      const int iy = ty * 16 - 1 + hy, ix = tx * 16 - 1 + hx;
      xs[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((int64_t)b * Cin + c) * H + iy) * W + ix] : 0.f;
    }
    for (int i = tid; i < 256 * Cout; i += 256) {
      const int p = i / Cout, nn = i - p * Cout;
      const int oy = ty * 16 + (p >> 4), ox = tx * 16 + (p & 15);
      dys[p * ldy + nn] = (oy < H && ox < W) ? ElemTraits<T>::ld(dy + (((int64_t)b * H + oy) * W + ox) * Cout + nn) : 0.f;
    }
    __syncthreads();
    for (int p = 0; p < 256; ++p) {
      const float g = dys[p * ldy + n];
      const int ly = p >> 4, lx = p & 15;
#pragma unroll
      for (int j = 0; j < MAXK; ++j) {
        const int k = grp + j * ngrp;
        if (k < KT) {
          const int c = k / 9, t = k - c * 9;
          acc[j] += g * xs[c * 324 + (ly + t / 3) * 18 + lx + t % 3];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXK; ++j) {
    const int k = grp + j * ngrp;
    if (k < KT) partial[((size_t)blockIdx.x * Cout + n) * KT + k] = acc[j];
  }
}

__global__ void sum_rows_flat_kernel(const float* partial, int rows, int64_t n, float* out, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int r = 0; r < rows; ++r) s += (double)partial[(size_t)r * n + i];
    out[i] = accumulate ? out[i] + (float)s : (float)s;
  }
}

// ------------------------------------------------------------------------------------------------
// optimisers (flat fp32 arena)
// ------------------------------------------------------------------------------------------------
__global__ void step_increment_kernel(int64_t* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

// 16-byte accesses (the arenas are 32-byte aligned; a scalar tail covers n % 4): the 4-byte version moved 2.1 TB/s
// DECOUPLED: torch.optim.AdamW (p *= 1 - lr * wd before the update); otherwise torch.optim.Adam (wd * p joins the gradient)
// MIRROR (round 5; SURVEY.md K8 "optional bf16 shadow write"): the updated parameter is also written as bf16 into `mirror` (the operand
// copy of the token GEMMs, plan_base.wb) -- 2 bytes per parameter more here instead of a cast pass that re-reads the fp32 arena
// (6 bytes per parameter, 0.21 ms per step at the 205 M parameters of FloodViT)
template <bool DECOUPLED, bool MIRROR = false>
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* step,
                            float lr, float b1, float b2, float eps, float wd, float gscale, bf16_t* mirror = nullptr) {
  const double t = (double)*step;
  const float bc1 = (float)(1.0 - pow((double)b1, t));
  const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
  const float step_size = lr / bc1;
  auto upd = [&](float gi, float& pi, float& mi, float& vi) {
    gi *= gscale;
    if (DECOUPLED) pi = pi * (1.f - lr * wd);
    else if (wd != 0.f) gi += wd * pi;
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + gi * gi * (1.f - b2);
    const float denom = sqrtf(vi) / bc2s + eps;
    pi = pi - step_size * (mi / denom);
  };
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const int64_t nv = vec ? n / 4 : 0;
  // streaming pass over four arenas that are far larger than L2 + MALL: non-temporal accesses, two vectors in flight per lane
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += 2 * stride) {
    const int64_t i2 = i + stride;
    const bool two = i2 < nv;
    const f32x4 ga = __builtin_nontemporal_load((const f32x4*)g + i);
    f32x4 pa = __builtin_nontemporal_load((f32x4*)p + i), ma = __builtin_nontemporal_load((f32x4*)m + i), va = __builtin_nontemporal_load((f32x4*)v + i);
    f32x4 gb = ga, pb = pa, mb = ma, vb = va;
    if (two) {
      gb = __builtin_nontemporal_load((const f32x4*)g + i2);
      pb = __builtin_nontemporal_load((f32x4*)p + i2); mb = __builtin_nontemporal_load((f32x4*)m + i2); vb = __builtin_nontemporal_load((f32x4*)v + i2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { float pj = pa[j], mj = ma[j], vj = va[j]; upd(ga[j], pj, mj, vj); pa[j] = pj; ma[j] = mj; va[j] = vj; }
    __builtin_nontemporal_store(pa, (f32x4*)p + i); __builtin_nontemporal_store(ma, (f32x4*)m + i); __builtin_nontemporal_store(va, (f32x4*)v + i);
    if constexpr (MIRROR) {
      uint2 w2;
      w2.x = (uint32_t)f32_to_bf16(pa[0]) | ((uint32_t)f32_to_bf16(pa[1]) << 16);
      w2.y = (uint32_t)f32_to_bf16(pa[2]) | ((uint32_t)f32_to_bf16(pa[3]) << 16);
      *((uint2*)mirror + i) = w2;
    }
    if (two) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { float pj = pb[j], mj = mb[j], vj = vb[j]; upd(gb[j], pj, mj, vj); pb[j] = pj; mb[j] = mj; vb[j] = vj; }
      __builtin_nontemporal_store(pb, (f32x4*)p + i2); __builtin_nontemporal_store(mb, (f32x4*)m + i2); __builtin_nontemporal_store(vb, (f32x4*)v + i2);
      if constexpr (MIRROR) {
        uint2 w2;
        w2.x = (uint32_t)f32_to_bf16(pb[0]) | ((uint32_t)f32_to_bf16(pb[1]) << 16);
        w2.y = (uint32_t)f32_to_bf16(pb[2]) | ((uint32_t)f32_to_bf16(pb[3]) << 16);
        *((uint2*)mirror + i2) = w2;
      }
    }
  }
  for (int64_t i = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(g[i], pi, mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if constexpr (MIRROR) mirror[i] = f32_to_bf16(pi);
  }
}

__global__ void sgd_kernel(float* p, const float* g, float* mom, int64_t n, const int64_t* step,
                           float lr, float mu, float wd, float gscale) {
  const bool first = *step <= 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    if (mu != 0.f) {
      const float b = first ? gi : mom[i] * mu + gi;
      mom[i] = b; gi = b;
    }
    p[i] = pi - lr * gi;
  }
}

// ------------------------------------------------------------------------------------------------
// layout helpers (tests only)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* x, T* y, int B, int C, int HW) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % C; int64_t r = i / C; const int p = r % HW; const int b = r / HW;
    ElemTraits<T>::st(y + i, x[((int64_t)b * C + c) * HW + p]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* x, float* y, int B, int C, int HW) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = i % HW; int64_t r = i / HW; const int c = r % C; const int b = r / C;
    y[i] = ElemTraits<T>::ld(x + ((int64_t)b * HW + p) * C + c);
  }
}

// MFMA fragment convention self-test: c[16][16] = a[16][KC] * b[16][KC]^T
template <typename T>
__global__ void selftest_mma_kernel(const T* a, const T* b, float* c) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int lane = threadIdx.x, g = lane >> 4, l15 = lane & 15;
  const u32x4 av = *(const u32x4*)(a + l15 * VEC * 4 + g * VEC);
  const u32x4 bv = *(const u32x4*)(b + l15 * VEC * 4 + g * VEC);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  mma16<T>(acc, av, bv);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[(g * 4 + r) * 16 + l15] = acc[r];
}

__global__ void selftest_tr16_kernel(const uint16_t* in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t buf[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) buf[i] = in[i];
  __syncthreads();
  const unsigned addr = (unsigned)(uintptr_t)buf + lane * 8;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)addr);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

int grid_for(int64_t n, int cap = 2048) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

bool chan_ok(int C, int dtype) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  return C % vec == 0 && C / vec <= kThreads / 2;
}

}  // namespace

// (bnfused.hip folds row lists longer than its workgroups sum themselves)
int ksmi_internal_fold_rows(float* partial, int rows, int K, int Cstride, int C, hipStream_t st) { return fold_rows(partial, rows, K, Cstride, C, st); }

// Column sums of a token matrix in ONE launch (bias gradients of nn.Linear: out[c] (+)= sum_r x[r][c], a few thousand rows):
// block = 4 adjacent 16-byte column vectors x 256 row lanes; fixed-order tree over the row lanes (deterministic).  The two-stage
// channel_sum + reduce_rows pair costs two launches (~21 us) per bias on the 3152-row FloodViT matrices.
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* x, int64_t rows, int C, float* out, int accumulate) {
  constexpr int VEC = ElemTraits<T>::kVec;
  __shared__ float red[256][4 * VEC + 1];
  const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = (blockIdx.x * 4 + q) * VEC;
  float a[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) a[j] = 0.f;
  if (c < C) {
    int64_t r = rl;
    for (; r + 3 * 256 < rows; r += 4 * 256) {                     // four rows in flight, summed in the order of the plain loop
      u32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *(const u32x4*)(x + (r + k * 256) * C + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f[VEC];
        vec_unpack<T>(v[k], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] += f[j];
      }
    }
    for (; r < rows; r += 256) {
      float f[VEC];
      vec_unpack<T>(*(const u32x4*)(x + r * C + c), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) a[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[rl][q * VEC + j] = a[j];
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (rl < s)
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[rl][q * VEC + j] += red[rl + s][q * VEC + j];
    __syncthreads();
  }
  if (rl == 0 && c < C)
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[c + j] = accumulate ? out[c + j] + red[0][q * VEC + j] : red[0][q * VEC + j];
}

#define KSMI_DT(dtype, EXPR_BF16, EXPR_F32)                                  \
  do {                                                                       \
    if ((dtype) == KSMI_BF16) { EXPR_BF16; }                                 \
    else if ((dtype) == KSMI_F32) { EXPR_F32; }                              \
    else return ksmi_fail(KSMI_E_ARG, "bad dtype");                          \
  } while (0)

extern "C" {

int ksmi_bn_finalize(const float* partial, int rows, int Cpad, int C, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, int training,
                     float* mean, float* rstd, float* scale, float* shift, void* stream) {
  if (C < 1 || (training && (!partial || rows < 1))) return ksmi_fail(KSMI_E_ARG, "bn_finalize: bad args");
  if (training) rows = fold_rows((float*)partial, rows, 2, Cpad, C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partial, rows, Cpad, C, count,
                     gamma, beta, running_mean, running_var, nbt, momentum, eps, training, mean, rstd, scale, shift);
  return ksmi_check_launch("bn_finalize");
}

int ksmi_reduce_rows_scaled(const float* partial, int rows, int K, int Cstride, int C, float* sums, float* dgamma, float* dbeta,
                            int accumulate, float alpha, void* stream) {
  if (rows < 1 || K < 1 || C < 1 || Cstride < C) return ksmi_fail(KSMI_E_ARG, "reduce_rows: bad args");
  rows = fold_rows((float*)partial, rows, K, Cstride, C, (hipStream_t)stream);
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partial, rows, K, Cstride, C, sums,
                     dgamma, dbeta, accumulate, alpha);
  return ksmi_check_launch("reduce_rows");
}
int ksmi_reduce_rows(const float* partial, int rows, int K, int Cstride, int C, float* sums, float* dgamma, float* dbeta,
                     int accumulate, void* stream) {
  return ksmi_reduce_rows_scaled(partial, rows, K, Cstride, C, sums, dgamma, dbeta, accumulate, 1.f, stream);
}

int ksmi_reduce_rows_batched(const ksmi_rowsum_desc* descs_device, int n, void* stream) {
  if (!descs_device || n < 1) return ksmi_fail(KSMI_E_ARG, "reduce_rows_batched: bad args");
  hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3(n, 32), dim3(256), 0, (hipStream_t)stream, descs_device);   // up to 512 channels
  return ksmi_check_launch("reduce_rows_batched");
}

int ksmi_reduce_rows_batched_wide(const ksmi_rowsum_desc* descs_device, int n, int max_c, void* stream) {
  if (!descs_device || n < 1 || max_c < 1 || (max_c + 15) / 16 > 65535) return ksmi_fail(KSMI_E_ARG, "reduce_rows_batched_wide: bad args");
  hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3(n, (max_c + 15) / 16), dim3(256), 0, (hipStream_t)stream, descs_device);
  return ksmi_check_launch("reduce_rows_batched_wide");
}

int ksmi_bn_add_relu(const void* z, const void* identity, const float* scale, const float* shift, void* out, int64_t npix,
                     int C, int dtype, void* stream) {
  if (!chan_ok(C, dtype)) return ksmi_fail(KSMI_E_ARG, "bn_add_relu: C must be a multiple of the 16-byte vector");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t nvec = npix * C / vec;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bn_add_relu_kernel<bf16_t>, dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                             (const bf16_t*)z, (const bf16_t*)identity, scale, shift, (bf16_t*)out, nvec, C / vec),
          hipLaunchKernelGGL(bn_add_relu_kernel<float>, dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                             (const float*)z, (const float*)identity, scale, shift, (float*)out, nvec, C / vec));
  return ksmi_check_launch("bn_add_relu");
}

int ksmi_bnrelu_bwd_reduce(const void* dout, const void* out, const void* z, const float* mean, const float* rstd,
                           float* partial, int rows, int64_t npix, int C, int dtype, void* stream) {
  if (!chan_ok(C, dtype) || rows < 1) return ksmi_fail(KSMI_E_ARG, "bnrelu_bwd_reduce: bad args");
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bnrelu_bwd_reduce_kernel<bf16_t>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                             (const bf16_t*)out, (const bf16_t*)z, mean, rstd, partial, npix, C),
          hipLaunchKernelGGL(bnrelu_bwd_reduce_kernel<float>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)dout,
                             (const float*)out, (const float*)z, mean, rstd, partial, npix, C));
  return ksmi_check_launch("bnrelu_bwd_reduce");
}

int ksmi_bnrelu_bwd_apply_scaled(void* dout_g, const void* out, const void* z, const float* mean, const float* rstd,
                                 const float* gamma, const float* sums, void* dz, double count, int64_t npix, int C, float alpha, int dtype,
                                 void* stream) {
  if (!chan_ok(C, dtype)) return ksmi_fail(KSMI_E_ARG, "bnrelu_bwd_apply: bad C");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t nvec = npix * C / vec;
  const float inv_n = (float)(1.0 / count);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<bf16_t>, dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                             (bf16_t*)dout_g, (const bf16_t*)out, (const bf16_t*)z, mean, rstd, gamma, sums, (bf16_t*)dz, inv_n, nvec,
                             C / vec, C, alpha),
          hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<float>, dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                             (float*)dout_g, (const float*)out, (const float*)z, mean, rstd, gamma, sums, (float*)dz, inv_n, nvec,
                             C / vec, C, alpha));
  return ksmi_check_launch("bnrelu_bwd_apply");
}
int ksmi_bnrelu_bwd_apply(void* dout_g, const void* out, const void* z, const float* mean, const float* rstd,
                          const float* gamma, const float* sums, void* dz, double count, int64_t npix, int C, int dtype,
                          void* stream) {
  return ksmi_bnrelu_bwd_apply_scaled(dout_g, out, z, mean, rstd, gamma, sums, dz, count, npix, C, 1.f, dtype, stream);
}

int ksmi_bn_bwd_apply_gated(const void* g, const void* z, const float* mean, const float* rstd, const float* gamma, const float* sums,
                            void* dz, double count, int64_t npix, int C, int dtype, void* stream) {
  if (!chan_ok(C, dtype)) return ksmi_fail(KSMI_E_ARG, "bn_bwd_apply_gated: bad C");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t nvec = npix * C / vec;
  const float inv_n = (float)(1.0 / count);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bn_bwd_apply_gated_kernel<bf16_t>, dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                             (const bf16_t*)g, (const bf16_t*)z, mean, rstd, gamma, sums, (bf16_t*)dz, inv_n, nvec, C / vec, C),
          hipLaunchKernelGGL(bn_bwd_apply_gated_kernel<float>, dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                             (const float*)g, (const float*)z, mean, rstd, gamma, sums, (float*)dz, inv_n, nvec, C / vec, C));
  return ksmi_check_launch("bn_bwd_apply_gated");
}

int ksmi_bn_bwd_apply_add(void* r_di, const void* g, const void* i, const float* mean, const float* rstd, const float* gamma,
                          const float* sums, float* partial, int rows, double count, int64_t npix, int C, int dtype,
                          void* stream) {
  if (!chan_ok(C, dtype) || rows < 1) return ksmi_fail(KSMI_E_ARG, "bn_bwd_apply_add: bad args");
  const float inv_n = (float)(1.0 / count);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bn_bwd_apply_add_kernel<bf16_t>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (bf16_t*)r_di,
                             (const bf16_t*)g, (const bf16_t*)i, mean, rstd, gamma, sums, partial, inv_n, npix, C),
          hipLaunchKernelGGL(bn_bwd_apply_add_kernel<float>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (float*)r_di,
                             (const float*)g, (const float*)i, mean, rstd, gamma, sums, partial, inv_n, npix, C));
  return ksmi_check_launch("bn_bwd_apply_add");
}

int ksmi_colsum(const void* x, int64_t rows, int C, float* out, int accumulate, int dtype, void* stream) {
  const int vec_ = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec_ || rows < 1 || !out) return ksmi_fail(KSMI_E_ARG, "colsum: bad args");
  const dim3 grid((C / vec_ + 3) / 4);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(1024), 0, (hipStream_t)stream, (const bf16_t*)x, rows, C, out, accumulate),
          hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(1024), 0, (hipStream_t)stream, (const float*)x, rows, C, out, accumulate));
  return ksmi_check_launch("colsum");
}

int ksmi_channel_sum(const void* x, float* partial, int rows, int64_t npix, int C, int dtype, void* stream) {
  const int vec_ = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec_ || rows < 1) return ksmi_fail(KSMI_E_ARG, "channel_sum: bad args");
  if (!chan_ok(C, dtype)) {
    const dim3 grid(rows, (C / vec_ + 63) / 64);
    KSMI_DT(dtype,
            hipLaunchKernelGGL(channel_sum_wide_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, partial, npix, C),
            hipLaunchKernelGGL(channel_sum_wide_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, partial, npix, C));
    return ksmi_check_launch("channel_sum_wide");
  }
  KSMI_DT(dtype,
          hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, partial, npix, C),
          hipLaunchKernelGGL(channel_sum_kernel<float>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)x, partial, npix, C));
  return ksmi_check_launch("channel_sum");
}

int ksmi_maxpool2x2_forward(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  if (!chan_ok(C, dtype) || (H & 1) || (W & 1)) return ksmi_fail(KSMI_E_ARG, "maxpool: even H,W and vector-multiple C required");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C),
          hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, B, H, W, C));
  return ksmi_check_launch("maxpool_fwd");
}

int ksmi_maxpool2x2_backward(const void* x, const void* dy, void* dx, int accumulate, int B, int H, int W, int C, int dtype,
                             void* stream) {
  if (!chan_ok(C, dtype) || (H & 1) || (W & 1)) return ksmi_fail(KSMI_E_ARG, "maxpool: even H,W and vector-multiple C required");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / vec);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, accumulate, B, H, W, C),
          hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, accumulate, B, H, W, C));
  return ksmi_check_launch("maxpool_bwd");
}

int ksmi_conv_first_stats_rows(int B, int H, int W) { return B * ((H + 15) / 16) * ((W + 15) / 16); }

static int first_conv_sources(const float* xtail, int* chead, int Cin, const float* nmean, const float* nstd, const float* nclamp) {
  if ((nmean != nullptr) != (nstd != nullptr) || (nmean != nullptr) != (nclamp != nullptr))
    return ksmi_fail(KSMI_E_ARG, "conv_first: mean, std and clamp come together");
  if (xtail == nullptr) *chead = Cin;
  else if (*chead < 1 || *chead >= Cin) return ksmi_fail(KSMI_E_ARG, "conv_first: a tail image needs 1 <= c_head < Cin");
  return 0;
}

int ksmi_conv_first_forward_raw(const float* x, const float* xtail, int chead, const float* w, const float* bias, void* out, float* stats,
                                int B, int Cin, int H, int W, int Cout, const float* nmean, const float* nstd, const float* nclamp,
                                int dtype, void* stream) {
  if (int rc = first_conv_sources(xtail, &chead, Cin, nmean, nstd, nclamp)) return rc;
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (Cin < 1 || Cin > 8 || Cout % vec || Cout > 256) return ksmi_fail(KSMI_E_ARG, "conv_first: Cin<=8, Cout multiple of vector, <=256");
  const int grid = ksmi_conv_first_stats_rows(B, H, W);
  const size_t lds = (size_t)(Cin * 324 + 256 * 17 + 512 + 16 * ((Cin * 9 + 3) / 4 * 4) + 16) * sizeof(float);    // conv_first_fwd_kernel: NC = 16
  hipStream_t st = (hipStream_t)stream;
#define KSMI_CF(CIN_)                                                                                                     \
  case CIN_:                                                                                                              \
    if (dtype == KSMI_BF16) hipLaunchKernelGGL((conv_first_fwd_kernel<bf16_t, CIN_>), dim3(grid), dim3(256), lds, st, x, w, bias, (bf16_t*)out, stats, B, H, W, Cout, nmean, nstd, nclamp, xtail, chead); \
    else hipLaunchKernelGGL((conv_first_fwd_kernel<float, CIN_>), dim3(grid), dim3(256), lds, st, x, w, bias, (float*)out, stats, B, H, W, Cout, nmean, nstd, nclamp, xtail, chead);                    \
    break;
  if (dtype != KSMI_BF16 && dtype != KSMI_F32) return ksmi_fail(KSMI_E_ARG, "bad dtype");
  switch (Cin) { KSMI_CF(1) KSMI_CF(2) KSMI_CF(3) KSMI_CF(4) KSMI_CF(5) KSMI_CF(6) KSMI_CF(7) KSMI_CF(8) }
#undef KSMI_CF
  return ksmi_check_launch("conv_first_fwd");
}

int ksmi_conv_first_forward(const float* x, const float* w, const float* bias, void* out, float* stats, int B, int Cin, int H,
                            int W, int Cout, int dtype, void* stream) {
  return ksmi_conv_first_forward_raw(x, nullptr, Cin, w, bias, out, stats, B, Cin, H, W, Cout, nullptr, nullptr, nullptr, dtype, stream);
}

static int conv_first_wgrad_blocks(int B, int H, int W) {
  const int patches = B * ((H + 15) / 16) * ((W + 15) / 16);
  return patches < 512 ? patches : 512;
}

size_t ksmi_conv_first_wgrad_workspace(int B, int Cin, int H, int W, int Cout) {
  return (size_t)conv_first_wgrad_blocks(B, H, W) * Cout * Cin * 9 * sizeof(float);
}

int ksmi_conv_first_wgrad(const float* x, const void* dy, float* dw, float* workspace, size_t ws_bytes, int B, int Cin, int H,
                          int W, int Cout, int accumulate, int dtype, void* stream) {
  if (Cin < 1 || Cin > 8 || 256 % Cout || (Cin * 9 + 256 / Cout - 1) / (256 / Cout) > 12)
    return ksmi_fail(KSMI_E_ARG, "conv_first_wgrad: unsupported Cin/Cout");
  if (ws_bytes < ksmi_conv_first_wgrad_workspace(B, Cin, H, W, Cout)) return ksmi_fail(KSMI_E_ARG, "conv_first_wgrad: workspace too small");
  const int blocks = conv_first_wgrad_blocks(B, H, W);
  const size_t lds = (size_t)(Cin * 324 + 256 * (Cout + 1)) * sizeof(float);
  KSMI_DT(dtype,
          {
            auto k = conv_first_wgrad_kernel<bf16_t>;
            if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, (const bf16_t*)dy, workspace, B, Cin, H, W, Cout);
          },
          {
            auto k = conv_first_wgrad_kernel<float>;
            if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, (hipStream_t)stream, x, (const float*)dy, workspace, B, Cin, H, W, Cout);
          });
  int rc = ksmi_check_launch("conv_first_wgrad");
  if (rc) return rc;
  const int64_t n = (int64_t)Cout * Cin * 9;
  hipLaunchKernelGGL(sum_rows_flat_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, workspace, blocks, n, dw, accumulate);
  return ksmi_check_launch("conv_first_wgrad_reduce");
}

int ksmi_im2col3x3_raw(const float* x_nchw, const float* xtail, int chead, void* out, int B, int Cin, int H, int W, int Kpad,
                       const float* nmean, const float* nstd, const float* nclamp, int dtype, void* stream) {
  if (int rc = first_conv_sources(xtail, &chead, Cin, nmean, nstd, nclamp)) return rc;
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (Cin < 1 || Cin * 9 > Kpad || Kpad % vec || Cin > 32) return ksmi_fail(KSMI_E_ARG, "im2col3x3: need Cin*9 <= Kpad, Kpad multiple of the vector");
  const int grid = B * ((H + 15) / 16) * ((W + 15) / 16);
  const size_t lds = (size_t)Cin * 324 * sizeof(float);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(im2col3x3_kernel<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, x_nchw, (bf16_t*)out, B, Cin, H, W, Kpad, nmean, nstd, nclamp, xtail, chead),
          hipLaunchKernelGGL(im2col3x3_kernel<float>, dim3(grid), dim3(256), lds, (hipStream_t)stream, x_nchw, (float*)out, B, Cin, H, W, Kpad, nmean, nstd, nclamp, xtail, chead));
  return ksmi_check_launch("im2col3x3");
}

int ksmi_im2col3x3(const float* x_nchw, void* out, int B, int Cin, int H, int W, int Kpad, int dtype, void* stream) {
  return ksmi_im2col3x3_raw(x_nchw, nullptr, Cin, out, B, Cin, H, W, Kpad, nullptr, nullptr, nullptr, dtype, stream);
}

int ksmi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t* step_count, float lr, float beta1,
                   float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !step_count || n < 0) return ksmi_fail(KSMI_E_ARG, "adam: bad args");
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_count);
  hipLaunchKernelGGL((adam_kernel<false, false>), dim3(grid_for((n + 3) / 4, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_count, lr, beta1,
                     beta2, eps, weight_decay, grad_scale, (bf16_t*)nullptr);
  return ksmi_check_launch("adam");
}

int ksmi_adam_step_mirror(float* p, const float* g, float* m, float* v, int64_t n, int64_t* step_count, float lr, float beta1,
                          float beta2, float eps, float weight_decay, float grad_scale, int decoupled, void* mirror_bf16, void* stream) {
  if (!p || !g || !m || !v || !step_count || !mirror_bf16 || n < 0 || ((uintptr_t)mirror_bf16 & 7)) return ksmi_fail(KSMI_E_ARG, "adam_mirror: bad args");
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_count);
  if (decoupled)
    hipLaunchKernelGGL((adam_kernel<true, true>), dim3(grid_for((n + 3) / 4, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_count, lr, beta1,
                       beta2, eps, weight_decay, grad_scale, (bf16_t*)mirror_bf16);
  else
    hipLaunchKernelGGL((adam_kernel<false, true>), dim3(grid_for((n + 3) / 4, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_count, lr, beta1,
                       beta2, eps, weight_decay, grad_scale, (bf16_t*)mirror_bf16);
  return ksmi_check_launch("adam_mirror");
}

int ksmi_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t* step_count, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !step_count || n < 0) return ksmi_fail(KSMI_E_ARG, "adamw: bad args");
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_count);
  hipLaunchKernelGGL((adam_kernel<true, false>), dim3(grid_for((n + 3) / 4, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_count, lr, beta1,
                     beta2, eps, weight_decay, grad_scale, (bf16_t*)nullptr);
  return ksmi_check_launch("adamw");
}

int ksmi_sgd_step(float* p, const float* g, float* mom, int64_t n, int64_t* step_count, float lr, float momentum,
                  float weight_decay, float grad_scale, void* stream) {
  if (!p || !g || !step_count || (momentum != 0.f && !mom)) return ksmi_fail(KSMI_E_ARG, "sgd: bad args");
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_count);
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, mom, n, step_count, lr, momentum,
                     weight_decay, grad_scale);
  return ksmi_check_launch("sgd");
}

int ksmi_fill_zero(void* p, size_t bytes, void* stream) {
  hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) return ksmi_fail((int)e, hipGetErrorString(e));
  return 0;
}

int ksmi_nchw_to_nhwc(const float* x, void* y, int B, int C, int HW, int dtype, void* stream) {
  const int64_t n = (int64_t)B * C * HW;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, B, C, HW),
          hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, (float*)y, B, C, HW));
  return ksmi_check_launch("nchw_to_nhwc");
}

int ksmi_nhwc_to_nchw(const void* x, float* y, int B, int C, int HW, int dtype, void* stream) {
  const int64_t n = (int64_t)B * C * HW;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, B, C, HW),
          hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, (const float*)x, y, B, C, HW));
  return ksmi_check_launch("nhwc_to_nchw");
}

int ksmi_selftest_mma(const void* a, const void* b, float* c, int dtype, void* stream) {
  KSMI_DT(dtype,
          hipLaunchKernelGGL(selftest_mma_kernel<bf16_t>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, c),
          hipLaunchKernelGGL(selftest_mma_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)a, (const float*)b, c));
  return ksmi_check_launch("selftest_mma");
}

int ksmi_selftest_tr16(const uint16_t* in256, uint16_t* out256, void* stream) {
  hipLaunchKernelGGL(selftest_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in256, out256);
  return ksmi_check_launch("selftest_tr16");
}

}  // extern "C"
