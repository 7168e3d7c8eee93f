// BatchNorm glue of conv_block_nested (models/snunet.py:19-29) with the statistics FINISH folded into the consumer (round 4).
//
// Until round 3 every train-mode BatchNorm call cost a latency-bound launch between the convolution that wrote the partial
// statistics rows and the pass that needs the finished statistics: ksmi_bn_finalize in the forward pass, ksmi_reduce_rows in the
// backward pass -- 76 launches of 2-32 workgroups per SNUNet step, each one a drain / 5 us kernel / ramp-up on the critical path.
// Here the consumer finishes the rows itself: the pass runs on <= 256 FAT workgroups (1024 threads, one per CU) that stride over the
// tensor, and every workgroup first sums the <= 512 partial rows (fp64, float4 loads, fixed order: all workgroups get the same
// bits), computes the per-channel constants into LDS, and then streams.  Workgroup 0 publishes what later launches read (saved
// mean / rstd / scale / shift, running statistics, dgamma / dbeta).  The arithmetic of the streaming part is the expression of the
// kernel it replaces, so the two paths agree bit for bit whenever the fp64 row sums round to the same float (tests:
// tests/test_gpu_kernels.py::test_bn_fused_*).
//
//   ksmi_bn_fin_add_relu           = ksmi_bn_finalize + ksmi_bn_add_relu (+ ksmi_maxpool2x2_forward)
//   ksmi_bn_bwd_fin_apply_gated    = ksmi_reduce_rows + ksmi_bn_bwd_apply_gated
//   ksmi_bnrelu_bwd_fin_apply      = ksmi_reduce_rows + ksmi_bnrelu_bwd_apply
//   ksmi_bn_bwd_fin_apply_add      = ksmi_reduce_rows + ksmi_bn_bwd_apply_add
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

int ksmi_internal_fold_rows(float* partial, int rows, int K, int Cstride, int C, hipStream_t st);   // elementwise.hip

namespace {

constexpr int kFat = 1024;     // threads per workgroup
constexpr int kMaxC = 512;     // channels the LDS tables hold
constexpr int kMaxRows = 512;  // rows a workgroup sums itself (longer lists are folded first)

// sums[k * C + c] = sum_r partial[(r * 2 + k) * Cstride + c], k < 2, fp64, identical in every workgroup.  C % 4 == 0, C <= kMaxC.
// Thread (column group g of 4 floats, row lane rl) walks rows rl, rl + lpc, ...; the row lanes are then added in order.
__device__ __forceinline__ void block_rows_sum(const float* __restrict__ partial, int rows, int Cstride, int C, double* red, double* sums) {
  const int Cq = C >> 2, G = Cq * 2;
  const int lpc = kFat / G;
  const int t = threadIdx.x;
  const int g = t % G, rl = t / G;
  if (rl < lpc) {
    const int k = g / Cq, c4 = g - k * Cq;
    const float* p = partial + (size_t)k * Cstride + c4 * 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
    for (int r = rl; r < rows; r += lpc) {
      const f32x4 v = *(const f32x4*)(p + (size_t)r * 2 * Cstride);
      a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
    double* o = red + ((size_t)rl * G + g) * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
  }
  __syncthreads();
  const int nl = lpc < rows ? lpc : rows;
  for (int v = t; v < 2 * C; v += kFat) {
    const int k = v / C, c = v - k * C;
    const int g2 = k * Cq + (c >> 2), j = c & 3;
    double s = 0.0;
    for (int r = 0; r < nl; ++r) s += red[((size_t)r * G + g2) * 4 + j];
    sums[v] = s;
  }
  __syncthreads();
}

struct FinArgs {
  const float* partial; int rows, Cstride, C; double count;
  const float* gamma; const float* beta; float* rmean; float* rvar; int64_t* nbt; float momentum, eps;
  float* mean_o; float* rstd_o; float* scale_o; float* shift_o;
};

// out = relu(z * scale + shift + identity) with (scale, shift) finished from the statistics rows of the convolution that wrote z.
// POOL: also y = maxpool2x2(out) (nn.MaxPool2d(2,2), models/snunet.py:73): one thread owns a 2x2 window.
template <typename T, bool POOL>
__global__ __launch_bounds__(kFat) void bn_fin_add_relu_kernel(FinArgs f, const T* __restrict__ z, const T* __restrict__ idn, T* __restrict__ out,
                                                               T* __restrict__ pooled, int B, int H, int W) {
  constexpr int VEC = ElemTraits<T>::kVec;
  __shared__ double red[kFat * 4];
  __shared__ double sums[2 * kMaxC];
  __shared__ float s_scale[kMaxC], s_shift[kMaxC];
  const int C = f.C;
  block_rows_sum(f.partial, f.rows, f.Cstride, C, red, sums);
  for (int c = threadIdx.x; c < C; c += kFat) {            // (the arithmetic of bn_finalize_kernel, elementwise.hip)
    const double mean = sums[c] / f.count;
    double var = sums[C + c] / f.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)f.eps));
    const float sc = f.gamma[c] * rs;
    const float sh = f.beta[c] - (float)mean * sc;
    s_scale[c] = sc; s_shift[c] = sh;
    if (blockIdx.x == 0) {
      f.mean_o[c] = (float)mean; f.rstd_o[c] = rs; f.scale_o[c] = sc; f.shift_o[c] = sh;
      const double unb = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
      f.rmean[c] = (1.f - f.momentum) * f.rmean[c] + f.momentum * (float)mean;
      f.rvar[c] = (1.f - f.momentum) * f.rvar[c] + f.momentum * (float)unb;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && f.nbt) *f.nbt += 1;
  __syncthreads();
  const int CV = C / VEC;
  const int64_t stride = (int64_t)gridDim.x * kFat;
  if (!POOL) {
    const int64_t nvec = (int64_t)B * H * W * CV;
    for (int64_t v = (int64_t)blockIdx.x * kFat + threadIdx.x; v < nvec; v += 2 * stride) {     // two vectors per trip: four loads in flight
      const int64_t v1 = v + stride;
      const bool two = v1 < nvec;
      const u32x4 rz0 = *(const u32x4*)(z + v * VEC), ri0 = *(const u32x4*)(idn + v * VEC);
      const u32x4 rz1 = *(const u32x4*)(z + (two ? v1 : v) * VEC), ri1 = *(const u32x4*)(idn + (two ? v1 : v) * VEC);
      float a[VEC], b[VEC];
      int c = (int)(v % CV) * VEC;
      vec_unpack<T>(rz0, a); vec_unpack<T>(ri0, b);
#pragma unroll
      for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j] * s_scale[c + j] + s_shift[c + j] + b[j], 0.f);
      *(u32x4*)(out + v * VEC) = vec_pack<T>(a);
      if (two) {
        c = (int)(v1 % CV) * VEC;
        vec_unpack<T>(rz1, a); vec_unpack<T>(ri1, b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j] * s_scale[c + j] + s_shift[c + j] + b[j], 0.f);
        *(u32x4*)(out + v1 * VEC) = vec_pack<T>(a);
      }
    }
  } else {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t n = (int64_t)B * Ho * Wo * CV;
    for (int64_t v = (int64_t)blockIdx.x * kFat + threadIdx.x; v < n; v += stride) {
      const int cv = (int)(v % CV); int64_t r = v / CV;
      const int ox = (int)(r % Wo); r /= Wo;
      const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
      const int64_t base = (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + cv * VEC;
      const int64_t offs[4] = {base, base + C, base + (int64_t)W * C, base + (int64_t)W * C + C};
      u32x4 rz[4], ri[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { rz[q] = *(const u32x4*)(z + offs[q]); ri[q] = *(const u32x4*)(idn + offs[q]); }
      const int c = cv * VEC;
      float m[VEC];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a[VEC], bb[VEC];
        vec_unpack<T>(rz[q], a); vec_unpack<T>(ri[q], bb);
#pragma unroll
        for (int j = 0; j < VEC; ++j) a[j] = fmaxf(a[j] * s_scale[c + j] + s_shift[c + j] + bb[j], 0.f);
        const u32x4 pk = vec_pack<T>(a);
        *(u32x4*)(out + offs[q]) = pk;
        vec_unpack<T>(pk, a);                       // the pooled value is the maximum of what the standalone pool would read
#pragma unroll
        for (int j = 0; j < VEC; ++j) m[j] = (q == 0 || a[j] > m[j]) ? a[j] : m[j];
      }
      *(u32x4*)(pooled + v * VEC) = vec_pack<T>(m);
    }
  }
}

struct BwdFinArgs {
  const float* partial; int rows, Cstride, C;
  float* sums_o; float* dgamma; float* dbeta; int accumulate;
  const float* mean; const float* rstd; const float* gamma;
};

// finishes (sum g, sum g * xhat) from the rows; s_sum[k * C + c] as floats (what ksmi_reduce_rows hands the apply kernels);
// workgroup 0 publishes them and the BatchNorm parameter gradients: dbeta (+)= sums[0], dgamma (+)= sums[1]
__device__ __forceinline__ void bwd_finish(const BwdFinArgs& f, double* red, double* sums, float* s_sum, float* s_mean, float* s_rstd, float* s_gamma) {
  const int C = f.C;
  block_rows_sum(f.partial, f.rows, f.Cstride, C, red, sums);
  for (int v = threadIdx.x; v < 2 * C; v += kFat) {
    const float s = (float)sums[v];
    s_sum[v] = s;
    if (blockIdx.x == 0) {
      if (f.sums_o) f.sums_o[v] = s;
      const int k = v / C, c = v - k * C;
      float* tgt = k == 0 ? f.dbeta : f.dgamma;
      if (tgt) tgt[c] = f.accumulate ? tgt[c] + 1.f * s : 1.f * s;
    }
  }
  for (int c = threadIdx.x; c < C; c += kFat) { s_mean[c] = f.mean[c]; s_rstd[c] = f.rstd[c]; s_gamma[c] = f.gamma[c]; }
  __syncthreads();
}

// MODE 0: g already gated (conv gate epilogue): dz = gamma * rstd * (g - s0/n - zhat * s1/n), g read only  (bn_bwd_apply_gated_kernel)
// MODE 1: g = dout * (out > 0) written over dout, then the same                                               (bnrelu_bwd_apply_kernel)
template <typename T, int MODE>
__global__ __launch_bounds__(kFat) void bn_bwd_fin_apply_kernel(BwdFinArgs f, T* __restrict__ g, const T* __restrict__ outp, const T* __restrict__ z,
                                                                T* __restrict__ dz, float inv_n, int64_t nvec) {
  constexpr int VEC = ElemTraits<T>::kVec;
  __shared__ double red[kFat * 4];
  __shared__ double sums[2 * kMaxC];
  __shared__ float s_sum[2 * kMaxC], s_mean[kMaxC], s_rstd[kMaxC], s_gamma[kMaxC];
  bwd_finish(f, red, sums, s_sum, s_mean, s_rstd, s_gamma);
  const int C = f.C, CV = C / VEC;
  const int64_t stride = (int64_t)gridDim.x * kFat;
  for (int64_t v = (int64_t)blockIdx.x * kFat + threadIdx.x; v < nvec; v += 2 * stride) {
    const int64_t v1 = v + stride;
    const bool two = v1 < nvec;
    const int64_t w1 = two ? v1 : v;
    const u32x4 rg0 = *(const u32x4*)(g + v * VEC), rz0 = *(const u32x4*)(z + v * VEC);
    const u32x4 rg1 = *(const u32x4*)(g + w1 * VEC), rz1 = *(const u32x4*)(z + w1 * VEC);
    u32x4 ro0, ro1;
    if (MODE == 1) { ro0 = *(const u32x4*)(outp + v * VEC); ro1 = *(const u32x4*)(outp + w1 * VEC); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const int64_t vv = u ? v1 : v;
      const int c = (int)(vv % CV) * VEC;
      float gg[VEC], zz[VEC], o[VEC];
      vec_unpack<T>(u ? rg1 : rg0, gg); vec_unpack<T>(u ? rz1 : rz0, zz);
      if (MODE == 1) vec_unpack<T>(u ? ro1 : ro0, o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (MODE == 1) gg[j] = o[j] > 0.f ? gg[j] : 0.f;
        const float zh = (zz[j] - s_mean[c + j]) * s_rstd[c + j];
        zz[j] = s_gamma[c + j] * s_rstd[c + j] * (gg[j] - s_sum[c + j] * inv_n - zh * s_sum[C + c + j] * inv_n);
      }
      if (MODE == 1) *(u32x4*)(g + vv * VEC) = vec_pack<T>(gg);
      *(u32x4*)(dz + vv * VEC) = vec_pack<T>(zz);
    }
  }
}

// di = g + gamma * rstd * (r - t0/n - xhat * t1/n) written over r; bias_partial[workgroup][c] = sum di  (bn_bwd_apply_add_kernel)
template <typename T>
__global__ __launch_bounds__(kFat) void bn_bwd_fin_apply_add_kernel(BwdFinArgs f, T* __restrict__ r, const T* __restrict__ g, const T* __restrict__ iv,
                                                                    float* __restrict__ bias_partial, float inv_n, int64_t npix) {
  constexpr int VEC = ElemTraits<T>::kVec;
  __shared__ double red[kFat * 4];
  __shared__ double sums[2 * kMaxC];
  __shared__ float s_sum[2 * kMaxC], s_mean[kMaxC], s_rstd[kMaxC], s_gamma[kMaxC];
  bwd_finish(f, red, sums, s_sum, s_mean, s_rstd, s_gamma);
  const int C = f.C, CV = C / VEC;
  const int npl = kFat / CV;                                     // pixel lanes (CV <= 256)
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const bool active = pl < npl;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (active) {
    float mu[VEC], rs[VEC], k0[VEC], k1[VEC], gr[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = cv * VEC + j;
      mu[j] = s_mean[c]; rs[j] = s_rstd[c]; gr[j] = s_gamma[c] * s_rstd[c];
      k0[j] = s_sum[c] * inv_n; k1[j] = s_sum[C + c] * inv_n;
    }
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = per * blockIdx.x, p1 = p0 + per < npix ? p0 + per : npix;
    for (int64_t p = p0 + pl; p < p1; p += 2 * npl) {
      const bool two = p + npl < p1;
      const int64_t offs[2] = {p * C + cv * VEC, two ? (p + npl) * C + cv * VEC : p * C + cv * VEC};
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
          rr[j] = gg[j] + gr[j] * (rr[j] - k0[j] - xh * k1[j]);
        }
        const u32x4 pk = vec_pack<T>(rr);
        *(u32x4*)(r + offs[u]) = pk;
        vec_unpack<T>(pk, rr);                 // the bias gradient sums what the next kernels will read
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += rr[j];
      }
    }
  }
  // pixel lanes -> one row per workgroup (reusing `red`: the statistics are in s_sum by now), fixed order
  float* fred = (float*)red;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VEC; ++j) fred[(size_t)threadIdx.x * VEC + j] = active ? acc[j] : 0.f;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kFat) {
    const int cvv = c / VEC, j = c - cvv * VEC;
    float s = 0.f;
    for (int l = 0; l < npl; ++l) s += fred[((size_t)l * CV + cvv) * VEC + j];
    bias_partial[(size_t)blockIdx.x * C + c] = s;
  }
}

bool fused_ok(int C, int dtype, int Cstride) {
  if (dtype != KSMI_BF16 && dtype != KSMI_F32) return false;
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  return C >= vec && C % vec == 0 && C % 4 == 0 && C <= kMaxC && Cstride >= C && Cstride % 4 == 0 && C / vec <= 256;
}

int fat_grid(int64_t work_items) {
  static const int cap = ksmi_knob_int("KSMI_BN_FAT_GRID", 256);     // (A/B switch: workgroups of the fused passes)
  int64_t b = (work_items + kFat - 1) / kFat;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

int ksmi_bn_fused_supported(int C, int Cstride, int dtype) { return fused_ok(C, dtype, Cstride) ? 1 : 0; }
int ksmi_bn_fused_max_rows(void) { return 256; }

int ksmi_bn_fin_add_relu(float* partial, int rows, int Cpad, int C, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                         float* mean, float* rstd, float* scale, float* shift, const void* z, const void* identity, void* out,
                         void* pooled, int B, int H, int W, int dtype, void* stream) {
  if (!fused_ok(C, dtype, Cpad) || !partial || rows < 1 || B < 1 || H < 1 || W < 1) return ksmi_fail(KSMI_E_ARG, "bn_fin_add_relu: bad args");
  if (pooled && ((H & 1) || (W & 1))) return ksmi_fail(KSMI_E_ARG, "bn_fin_add_relu: pooling needs even H, W");
  if (rows > kMaxRows) rows = ksmi_internal_fold_rows(partial, rows, 2, Cpad, C, (hipStream_t)stream);
  FinArgs f{partial, rows, Cpad, C, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, mean, rstd, scale, shift};
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t items = pooled ? (int64_t)B * (H / 2) * (W / 2) * (C / vec) : ((int64_t)B * H * W * (C / vec) + 1) / 2;
  const dim3 grid(fat_grid(items));
#define KSMI_FAR(TT, PP) hipLaunchKernelGGL((bn_fin_add_relu_kernel<TT, PP>), grid, dim3(kFat), 0, (hipStream_t)stream, f, (const TT*)z, \
                                            (const TT*)identity, (TT*)out, (TT*)pooled, B, H, W)
  if (dtype == KSMI_BF16) { if (pooled) KSMI_FAR(bf16_t, true); else KSMI_FAR(bf16_t, false); }
  else { if (pooled) KSMI_FAR(float, true); else KSMI_FAR(float, false); }
#undef KSMI_FAR
  return ksmi_check_launch("bn_fin_add_relu");
}

static int bwd_apply(int mode, float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                     void* g, const void* outp, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
                     double count, int64_t npix, int C, int dtype, void* stream, const char* what) {
  if (!fused_ok(C, dtype, Cstride) || !partial || rows < 1 || npix < 1) return ksmi_fail(KSMI_E_ARG, what);
  if (rows > kMaxRows) rows = ksmi_internal_fold_rows(partial, rows, 2, Cstride, C, (hipStream_t)stream);
  BwdFinArgs f{partial, rows, Cstride, C, sums, dgamma, dbeta, accumulate, mean, rstd, gamma};
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  const int64_t nvec = npix * C / vec;
  const float inv_n = (float)(1.0 / count);
  const dim3 grid(fat_grid((nvec + 1) / 2));
#define KSMI_BFA(TT, MM) hipLaunchKernelGGL((bn_bwd_fin_apply_kernel<TT, MM>), grid, dim3(kFat), 0, (hipStream_t)stream, f, (TT*)g, \
                                            (const TT*)outp, (const TT*)z, (TT*)dz, inv_n, nvec)
  if (dtype == KSMI_BF16) { if (mode) KSMI_BFA(bf16_t, 1); else KSMI_BFA(bf16_t, 0); }
  else { if (mode) KSMI_BFA(float, 1); else KSMI_BFA(float, 0); }
#undef KSMI_BFA
  return ksmi_check_launch(what);
}

int ksmi_bn_bwd_fin_apply_gated(float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                                const void* g, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
                                double count, int64_t npix, int C, int dtype, void* stream) {
  return bwd_apply(0, partial, rows, Cstride, sums, dgamma, dbeta, accumulate, (void*)g, nullptr, z, mean, rstd, gamma, dz, count, npix, C,
                   dtype, stream, "bn_bwd_fin_apply_gated");
}

int ksmi_bnrelu_bwd_fin_apply(float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                              void* dout_g, const void* out, const void* z, const float* mean, const float* rstd, const float* gamma,
                              void* dz, double count, int64_t npix, int C, int dtype, void* stream) {
  return bwd_apply(1, partial, rows, Cstride, sums, dgamma, dbeta, accumulate, dout_g, out, z, mean, rstd, gamma, dz, count, npix, C, dtype,
                   stream, "bnrelu_bwd_fin_apply");
}

int ksmi_bn_bwd_fin_apply_add(float* partial, int rows, int Cstride, float* sums, float* dgamma, float* dbeta, int accumulate,
                              void* r_di, const void* g, const void* i, const float* mean, const float* rstd, const float* gamma,
                              float* bias_partial, int bias_rows, double count, int64_t npix, int C, int dtype, void* stream) {
  if (!fused_ok(C, dtype, Cstride) || !partial || rows < 1 || npix < 1 || bias_rows < 1 || bias_rows > 256 || !bias_partial)
    return ksmi_fail(KSMI_E_ARG, "bn_bwd_fin_apply_add: bad args");
  if (rows > kMaxRows) rows = ksmi_internal_fold_rows(partial, rows, 2, Cstride, C, (hipStream_t)stream);
  BwdFinArgs f{partial, rows, Cstride, C, sums, dgamma, dbeta, accumulate, mean, rstd, gamma};
  const float inv_n = (float)(1.0 / count);
  if (dtype == KSMI_BF16)
    hipLaunchKernelGGL(bn_bwd_fin_apply_add_kernel<bf16_t>, dim3(bias_rows), dim3(kFat), 0, (hipStream_t)stream, f, (bf16_t*)r_di,
                       (const bf16_t*)g, (const bf16_t*)i, bias_partial, inv_n, npix);
  else
    hipLaunchKernelGGL(bn_bwd_fin_apply_add_kernel<float>, dim3(bias_rows), dim3(kFat), 0, (hipStream_t)stream, f, (float*)r_di,
                       (const float*)g, (const float*)i, bias_partial, inv_n, npix);
  return ksmi_check_launch("bn_bwd_fin_apply_add");
}

}  // extern "C"
