// Stochastic layers of ChangeFormerV6 (changeformer.py:107,126-132 Mlp.drop; :160-162,203-207 attn_drop / proj_drop;
// :236-241 DropPath on both residual branches): y = resid + DropPath_b(Dropout(x)).
// Masks are never stored: every draw is ksmi_rng_keep(key(seed, step, site), element index) (common.h), so the backward
// pass calls the same kernel on the gradient and the CPU oracle regenerates the masks bit for bit (oracle/rng_ref.py).
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void drop_apply_kernel(const T* x, const T* resid, T* y, int64_t nvec,   // y may alias x
                                                         int cols, int rows_per_sample, uint32_t thr, float inv, uint32_t site,
                                                         uint32_t dp_thr, float dp_inv, uint32_t dp_site, const uint32_t* __restrict__ state) {
  constexpr int V = ElemTraits<T>::kVec;
  const uint32_t key = thr ? ksmi_rng_key(state, site) : 0u;
  const uint32_t dkey = dp_thr ? ksmi_rng_key(state, dp_site) : 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t e0 = (uint32_t)(i * V);
    float f[V];
    vec_unpack<T>(((const u32x4*)x)[i], f);
    float s = 1.f;
    if (dp_thr) {
      const uint32_t sample = (e0 / (uint32_t)cols) / (uint32_t)rows_per_sample;
      s = ksmi_rng_keep(dkey, sample, dp_thr) ? dp_inv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float m = s;
      if (thr) m = ksmi_rng_keep(key, e0 + j, thr) ? inv * s : 0.f;
      f[j] *= m;
    }
    if (resid) {
      float r[V];
      vec_unpack<T>(((const u32x4*)resid)[i], r);
#pragma unroll
      for (int j = 0; j < V; ++j) f[j] += r[j];
    }
    ((u32x4*)y)[i] = vec_pack<T>(f);
  }
}

__global__ void rng_advance_kernel(uint32_t* state) { state[1] += 1u; }

// y = relu(z * scale[c] + shift[c]) * Dropout2d(b, c): nn.Dropout2d zeroes whole (sample, channel) planes (siam_conc.py:20 ...)
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_drop2d_kernel(const T* z, const float* __restrict__ scale, const float* __restrict__ shift, T* y,
                                                             int64_t nvec, int CV, int64_t hw_cv, uint32_t thr, float inv, uint32_t site,
                                                             const uint32_t* __restrict__ state) {
  constexpr int V = ElemTraits<T>::kVec;
  const uint32_t key = thr ? ksmi_rng_key(state, site) : 0u;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % CV) * V;
    const uint32_t b = (uint32_t)(i / hw_cv);
    float f[V];
    vec_unpack<T>(((const u32x4*)z)[i], f);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float v = fmaxf(f[j] * scale[c + j] + shift[c + j], 0.f);
      if (thr) v = ksmi_rng_keep(key, b * (uint32_t)(CV * V) + c + j, thr) ? v * inv : 0.f;
      f[j] = v;
    }
    ((u32x4*)y)[i] = vec_pack<T>(f);
  }
}

// FC-Siam-diff skip (siam_diff.py:137,146,154,161): y = |a - b|; adjoint: da = dy * sign(a - b), db = -da (da/db (+)=)
template <typename T>
__global__ __launch_bounds__(256) void absdiff_kernel(const T* a, const T* b, T* y, int64_t nvec) {
  constexpr int V = ElemTraits<T>::kVec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float fa[V], fb[V];
    vec_unpack<T>(((const u32x4*)a)[i], fa);
    vec_unpack<T>(((const u32x4*)b)[i], fb);
#pragma unroll
    for (int j = 0; j < V; ++j) fa[j] = fabsf(fa[j] - fb[j]);
    ((u32x4*)y)[i] = vec_pack<T>(fa);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void absdiff_bwd_kernel(const T* a, const T* b, const T* dy, T* da, T* db, int acc_a, int acc_b, int64_t nvec) {
  constexpr int V = ElemTraits<T>::kVec;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float fa[V], fb[V], g[V], oa[V], ob[V];
    vec_unpack<T>(((const u32x4*)a)[i], fa);
    vec_unpack<T>(((const u32x4*)b)[i], fb);
    vec_unpack<T>(((const u32x4*)dy)[i], g);
    if (acc_a) vec_unpack<T>(((const u32x4*)da)[i], oa);
    if (acc_b) vec_unpack<T>(((const u32x4*)db)[i], ob);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float d = fa[j] - fb[j];
      const float s = d > 0.f ? g[j] : (d < 0.f ? -g[j] : 0.f);     // torch.abs: subgradient 0 at 0
      oa[j] = acc_a ? oa[j] + s : s;
      ob[j] = acc_b ? ob[j] - s : -s;
    }
    ((u32x4*)da)[i] = vec_pack<T>(oa);
    ((u32x4*)db)[i] = vec_pack<T>(ob);
  }
}

}  // namespace

int ksmi_rng_advance(uint32_t* state, void* stream) {
  if (!state) return ksmi_fail(KSMI_E_ARG, "rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  return ksmi_check_launch("rng_advance");
}

int ksmi_dropout_apply(const void* x, const void* resid, void* y, int64_t rows, int cols, int rows_per_sample, uint32_t thr, float inv_keep,
                       uint32_t site, uint32_t dp_thr, float dp_inv_keep, uint32_t dp_site, const uint32_t* rng_state, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (dtype != KSMI_BF16 && dtype != KSMI_F32) return ksmi_fail(KSMI_E_ARG, "dropout_apply: dtype");
  if (cols <= 0 || cols % vec) return ksmi_fail(KSMI_E_ARG, "dropout_apply: cols must be a multiple of the 16-byte vector");
  if ((thr || dp_thr) && !rng_state) return ksmi_fail(KSMI_E_ARG, "dropout_apply: the rng state is required");
  if (dp_thr && rows_per_sample <= 0) return ksmi_fail(KSMI_E_ARG, "dropout_apply: rows_per_sample");
  if ((uint64_t)rows * cols >= (1ull << 32)) return ksmi_fail(KSMI_E_UNSUPPORTED, "dropout_apply: rows*cols must stay below 2^32");
  const int64_t nvec = rows * cols / vec;
  if (nvec == 0) return 0;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == KSMI_BF16)
    hipLaunchKernelGGL(drop_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)resid, (bf16_t*)y, nvec,
                       cols, rows_per_sample, thr, inv_keep, site, dp_thr, dp_inv_keep, dp_site, rng_state);
  else
    hipLaunchKernelGGL(drop_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (const float*)resid, (float*)y, nvec,
                       cols, rows_per_sample, thr, inv_keep, site, dp_thr, dp_inv_keep, dp_site, rng_state);
  return ksmi_check_launch("dropout_apply");
}

int ksmi_bn_relu_drop2d(const void* z, const float* scale, const float* shift, void* y, int B, int64_t HW, int C, uint32_t thr, float inv_keep,
                        uint32_t site, const uint32_t* rng_state, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (dtype != KSMI_BF16 && dtype != KSMI_F32) return ksmi_fail(KSMI_E_ARG, "bn_relu_drop2d: dtype");
  if (C <= 0 || C % vec) return ksmi_fail(KSMI_E_ARG, "bn_relu_drop2d: C must be a multiple of the 16-byte vector");
  if (thr && !rng_state) return ksmi_fail(KSMI_E_ARG, "bn_relu_drop2d: the rng state is required");
  const int64_t nvec = (int64_t)B * HW * (C / vec);
  if (nvec == 0) return 0;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == KSMI_BF16)
    hipLaunchKernelGGL(bn_relu_drop2d_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)z, scale, shift, (bf16_t*)y, nvec, C / vec,
                       HW * (C / vec), thr, inv_keep, site, rng_state);
  else
    hipLaunchKernelGGL(bn_relu_drop2d_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)z, scale, shift, (float*)y, nvec, C / vec,
                       HW * (C / vec), thr, inv_keep, site, rng_state);
  return ksmi_check_launch("bn_relu_drop2d");
}

int ksmi_absdiff_forward(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (n % vec) return ksmi_fail(KSMI_E_ARG, "absdiff: element count must be a multiple of the 16-byte vector");
  const int64_t nvec = n / vec;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == KSMI_BF16) hipLaunchKernelGGL(absdiff_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, nvec);
  else hipLaunchKernelGGL(absdiff_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)y, nvec);
  return ksmi_check_launch("absdiff");
}

int ksmi_absdiff_backward(const void* a, const void* b, const void* dy, void* da, void* db, int accumulate_a, int accumulate_b, int64_t n,
                          int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (n % vec) return ksmi_fail(KSMI_E_ARG, "absdiff_bwd: element count must be a multiple of the 16-byte vector");
  const int64_t nvec = n / vec;
  int64_t blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == KSMI_BF16)
    hipLaunchKernelGGL(absdiff_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)dy,
                       (bf16_t*)da, (bf16_t*)db, accumulate_a, accumulate_b, nvec);
  else
    hipLaunchKernelGGL(absdiff_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)dy,
                       (float*)da, (float*)db, accumulate_a, accumulate_b, nvec);
  return ksmi_check_launch("absdiff_bwd");
}
