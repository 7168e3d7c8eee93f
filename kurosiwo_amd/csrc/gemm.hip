// Token GEMMs (bf16) of the transformer paths: every nn.Linear of FloodViT (vision_transformer.py:22-31,47-50) and ChangeFormer
// (changeformer.py:110-113,157-161,141), forward and input gradient, on 128 x (64|128) output tiles:
//   gemm_nt : Y[m][n]  = sum_k X[m][k]  W[n][k] + bias[n] (+ R[m][n])          (forward)
//   gemm_nn : dX[m][k] = sum_n dY[m][n] W[n][k]           (+= optional)        (input gradient)
// W is the bf16 mirror of the fp32 parameter arena (one cast launch per step), row-major [N][K] as nn.Linear stores it, so the
// forward reads it K-contiguous (plain 16-byte LDS fragment reads) and the input gradient reads it transposed from the same
// row-major LDS image with ds_read_b64_tr_b16.  Swapped-operand MFMA (D = W X^T) + permuted tile rows: a lane owns 16
// consecutive output channels of one token row -> 32-byte contiguous stores.  4 waves (2 x 2), K steps of 32, double-buffered
// LDS with register prefetch.  (The implicit-GEMM kernel gave these shapes 215 TFLOP/s: 256 x 32 tiles re-stage the activations
// per 32 output channels.)
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

__device__ __forceinline__ int swz4(int row) { return (0 - (row >> 2)) & 3; }
__device__ __forceinline__ u32x4 ldv(const bf16_t* p, bool ok) { return ok ? *(const u32x4*)p : (u32x4){0u, 0u, 0u, 0u}; }

struct GemmP {
  const bf16_t* a; int a_rs;          // activations [rows][..]
  const bf16_t* w; int w_rs;          // weights [N][K] bf16
  const float* bias; const bf16_t* resid; int r_rs;
  bf16_t* out; int o_rs;
  int rows, K, N, accumulate;
};

// lane's 16 consecutive outputs (acc[t][.][r], t = 0..3) of row m -> two 16-byte stores
template <int MT>
__device__ __forceinline__ void store_row(const GemmP& p, const f32x4 (&acc)[4][MT], int mt, int m, int c0, int ncols, bool has_bias, const float* bias16) {
  if (m >= p.rows) return;
  float v[16];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][mt][r] + (has_bias ? bias16[t * 4 + r] : 0.f);
  bf16_t* op = p.out + (size_t)m * p.o_rs + c0;
  if (c0 + 16 <= ncols) {
    if (p.resid) {
      float a[8], b[8];
      vec_unpack<bf16_t>(*(const u32x4*)(p.resid + (size_t)m * p.r_rs + c0), a);
      vec_unpack<bf16_t>(*(const u32x4*)(p.resid + (size_t)m * p.r_rs + c0 + 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] += a[j]; v[8 + j] += b[j]; }
    }
    if (p.accumulate) {
      float a[8], b[8];
      vec_unpack<bf16_t>(*(const u32x4*)op, a);
      vec_unpack<bf16_t>(*(const u32x4*)(op + 8), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] += a[j]; v[8 + j] += b[j]; }
    }
    *(u32x4*)op = vec_pack<bf16_t>(v);
    *(u32x4*)(op + 8) = vec_pack<bf16_t>(v + 8);
  } else {
    for (int j = 0; j < 16 && c0 + j < ncols; ++j) {
      float o = v[j];
      if (p.resid) o += bf16_to_f32(p.resid[(size_t)m * p.r_rs + c0 + j]);
      if (p.accumulate) o += bf16_to_f32(op[j]);
      op[j] = f32_to_bf16(o);
    }
  }
}

// ---------------------------------------------------------------------------------------------- forward (NT)
// WG tile: 128 rows x TN output channels.  TN = 128: wave (wn, wm) = 64 channels x 64 rows (4 x 4 MFMA tiles); TN = 64 (grids that
// would not fill the chip with 128-wide tiles): wave wm = 64 channels x 32 rows (4 x 2 tiles)
template <int TN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmP p) {
  constexpr int TM = 128, KS = 32, MT = TN == 128 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (TM + TN) * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int wn = TN == 128 ? wave >> 1 : 0, wm = TN == 128 ? wave & 1 : wave;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  const int nsteps = (p.K + KS - 1) / KS;
  constexpr int WV = TN / 64;                                  // W-tile vectors per thread
  u32x4 rx[2], rw[WV];
  auto gload = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + i * 256, r = v >> 2, q = v & 3;
      const int k = s * KS + q * 8;
      rx[i] = ldv(p.a + (size_t)(m0 + r) * p.a_rs + k, m0 + r < p.rows && k < p.K);
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256, r = v >> 2, q = v & 3;
      const int k = s * KS + q * 8;
      // LDS row r (= MFMA row index nt*16 + g*4 + rr within the 64-row half) holds channel g*16 + nt*4 + rr
      const int h = r >> 6, j = r & 63;
      const int nloc = h * 64 + ((j >> 2) & 3) * 16 + (j >> 4) * 4 + (j & 3);
      rw[i] = ldv(p.w + (size_t)(n0 + nloc) * p.w_rs + k, n0 + nloc < p.N && k < p.K);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* bx = smem + buf * (TM + TN) * 64;
    unsigned char* bw = bx + TM * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + i * 256, r = v >> 2, q = v & 3;
      *(u32x4*)(bx + r * 64 + ((q ^ swz4(r)) << 4)) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256, r = v >> 2, q = v & 3;
      *(u32x4*)(bw + r * 64 + ((q ^ swz4(r)) << 4)) = rw[i];
    }
  };
  f32x4 acc[4][MT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  gload(0); lstore(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) gload(s + 1);
    const unsigned char* bx = smem + (s & 1) * (TM + TN) * 64;
    const unsigned char* bw = bx + TM * 64;
    u32x4 fx[MT], fw[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rw_ = wn * 64 + t * 16 + l15;
      fw[t] = *(const u32x4*)(bw + rw_ * 64 + ((g ^ swz4(rw_)) << 4));
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int rx_ = wm * 16 * MT + t * 16 + l15;
      fx[t] = *(const u32x4*)(bx + rx_ * 64 + ((g ^ swz4(rx_)) << 4));
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) mma16<bf16_t>(acc[a][b], fw[a], fx[b]);    // rows = channels, cols = token rows
    if (more) lstore((s + 1) & 1);
    __syncthreads();
  }
  const int c0 = n0 + wn * 64 + g * 16;
  float bias16[16];
  const bool has_bias = p.bias != nullptr;
  if (has_bias)
#pragma unroll
    for (int j = 0; j < 16; ++j) bias16[j] = c0 + j < p.N ? p.bias[c0 + j] : 0.f;
  if (c0 >= p.N) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) store_row<MT>(p, acc, mt, m0 + wm * 16 * MT + mt * 16 + l15, c0, p.N, has_bias, bias16);
}

// ---------------------------------------------------------------------------------------------- input gradient (NN)
// out[m][k] = sum_n a[m][n] w[n][k]; p.K = number of output columns (k), p.N = reduction length (n)
template <int TK>
__global__ __launch_bounds__(256) void gemm_nn_kernel(GemmP p) {
  constexpr int TM = 128, NS = 32, MT = TK == 128 ? 4 : 2, WROW = TK * 2, WVR = TK / 8;     // W tile: 32 rows of WROW bytes
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (TM * 64 + NS * TK * 2)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int wk = TK == 128 ? wave >> 1 : 0, wm = TK == 128 ? wave & 1 : wave;
  const int m0 = blockIdx.x * TM, k0 = blockIdx.y * TK;
  const int nsteps = (p.N + NS - 1) / NS;
  constexpr int WV = TK / 64, GM = WROW / 32 - 1;            // W-tile vectors per thread; granule mask of the swizzle
  u32x4 ry[2], rw[WV];
  auto gload = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + i * 256;
      const int r = v >> 2, q = v & 3;                       // dY tile: 128 rows x 4 vectors
      const int n = s * NS + q * 8;
      ry[i] = ldv(p.a + (size_t)(m0 + r) * p.a_rs + n, m0 + r < p.rows && n < p.N);
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256;
      const int wr = v / WVR, wq = v - wr * WVR;             // W tile: 32 rows (n) x WVR vectors (k)
      rw[i] = ldv(p.w + (size_t)(s * NS + wr) * p.w_rs + k0 + wq * 8, s * NS + wr < p.N && k0 + wq * 8 < p.K);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* by = smem + buf * (TM * 64 + NS * TK * 2);
    unsigned char* bw = by + TM * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + i * 256;
      const int r = v >> 2, q = v & 3;
      *(u32x4*)(by + r * 64 + ((q ^ swz4(r)) << 4)) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256;
      const int wr = v / WVR, wq = v - wr * WVR;
      *(u32x4*)(bw + wr * WROW + (((((wq * 16) >> 5) ^ (wr & 7)) & GM) << 5) + ((wq * 16) & 31)) = rw[i];   // 32-byte granule swizzle
    }
  };
  f32x4 acc[4][MT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  gload(0); lstore(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) gload(s + 1);
    const unsigned char* by = smem + (s & 1) * (TM * 64 + NS * TK * 2);
    const unsigned bw = (unsigned)(uintptr_t)(by + TM * 64);
    u32x4 fy[MT], fw[4];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int ry_ = wm * 16 * MT + t * 16 + l15;
      fy[t] = *(const u32x4*)(by + ry_ * 64 + ((g ^ swz4(ry_)) << 4));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // W^T fragment, output-column tile t: lane i = l15 is MFMA row i <-> output column wk*64 + (i>>2)*16 + t*4 + (i&3);
      // the lane's 4-column chunk (q = l15 & 3) of LDS row (n = g*8 + jr [+4]) starts at column wk*64 + q*16 + t*4
      const int jr = l15 >> 2, q = l15 & 3;
      const int cb = (wk * 64 + q * 16 + t * 4) * 2;
      const int r0 = g * 8 + jr, r1 = r0 + 4;
      const unsigned a0 = bw + r0 * WROW + ((((cb >> 5) ^ (r0 & 7)) & GM) << 5) + (cb & 31);
      const unsigned a1 = bw + r1 * WROW + ((((cb >> 5) ^ (r1 & 7)) & GM) << 5) + (cb & 31);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
      fw[t][0] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
      fw[t][1] = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
      fw[t][2] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
      fw[t][3] = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) mma16<bf16_t>(acc[a][b], fw[a], fy[b]);    // rows = output columns, cols = token rows
    if (more) lstore((s + 1) & 1);
    __syncthreads();
  }
  const int c0 = k0 + wk * 64 + g * 16;
  if (c0 >= p.K) return;
  float nob[16];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) store_row<MT>(p, acc, mt, m0 + wm * 16 * MT + mt * 16 + l15, c0, p.K, false, nob);
}

__global__ void cast_bf16_kernel(const float* src, bf16_t* dst, int64_t nvec) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 a = *(const f32x4*)(src + v * 8), b = *(const f32x4*)(src + v * 8 + 4);
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(u32x4*)(dst + v * 8) = vec_pack<bf16_t>(f);
  }
}

}  // namespace

extern "C" {

int ksmi_cast_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (n % 8) return ksmi_fail(KSMI_E_ARG, "cast_bf16: element count must be a multiple of 8");
  const int64_t nvec = n / 8;
  int64_t blocks = (nvec + 255) / 256; if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, nvec);
  return ksmi_check_launch("cast_bf16");
}

static int gemm_args_ok(int a_rs, int w_rs, int o_rs, int r_rs, int K, int N) {
  return ((a_rs | w_rs | o_rs | r_rs | K | N) & 7) == 0;
}

int ksmi_gemm_nt(const void* x, int x_rs, const void* w, int w_rs, const float* bias, const void* resid, int r_rs, void* y, int y_rs,
                 int rows, int K, int N, void* stream) {
  if (!gemm_args_ok(x_rs, w_rs, y_rs, r_rs, K, N) || rows < 1) return ksmi_fail(KSMI_E_ARG, "gemm_nt: strides, K and N must be multiples of 8");
  { const int r2 = ksmi_gemm2_nt(x, x_rs, w, w_rs, bias, resid, r_rs, y, y_rs, rows, K, N, (hipStream_t)stream); if (r2 <= 0) return r2; }
  GemmP p = {(const bf16_t*)x, x_rs, (const bf16_t*)w, w_rs, bias, (const bf16_t*)resid, r_rs, (bf16_t*)y, y_rs, rows, K, N, 0};
  const int mtiles = (rows + 127) / 128;
  if (mtiles * ((N + 127) / 128) >= 512) hipLaunchKernelGGL(gemm_nt_kernel<128>, dim3(mtiles, (N + 127) / 128), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_nt_kernel<64>, dim3(mtiles, (N + 63) / 64), dim3(256), 0, (hipStream_t)stream, p);
  return ksmi_check_launch("gemm_nt");
}

int ksmi_gemm_nn(const void* dy, int dy_rs, const void* w, int w_rs, void* dx, int dx_rs, int rows, int K, int N, int accumulate, void* stream) {
  if (!gemm_args_ok(dy_rs, w_rs, dx_rs, 0, K, N) || rows < 1) return ksmi_fail(KSMI_E_ARG, "gemm_nn: strides, K and N must be multiples of 8");
  { const int r2 = ksmi_gemm2_nn(dy, dy_rs, w, w_rs, dx, dx_rs, rows, K, N, accumulate, (hipStream_t)stream); if (r2 <= 0) return r2; }
  GemmP p = {(const bf16_t*)dy, dy_rs, (const bf16_t*)w, w_rs, nullptr, nullptr, 0, (bf16_t*)dx, dx_rs, rows, K, N, accumulate};
  const int mtiles = (rows + 127) / 128;
  if (mtiles * ((K + 127) / 128) >= 512) hipLaunchKernelGGL(gemm_nn_kernel<128>, dim3(mtiles, (K + 127) / 128), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_nn_kernel<64>, dim3(mtiles, (K + 63) / 64), dim3(256), 0, (hipStream_t)stream, p);
  return ksmi_check_launch("gemm_nn");
}

}  // extern "C"
