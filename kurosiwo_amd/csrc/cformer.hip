// ChangeFormerV6 glue kernels (rows C1-C7 of SURVEY.md §8(a)); reference: models/changeformer.py.
// Activations are NHWC / token-major [rows][C] in T (bf16 / fp32).  The dense contractions (patch-embed and
// spatial-reduction convolutions via im2col, every nn.Linear, the 3x3 / 1x1 / transposed convolutions of the decoder)
// run on the implicit-GEMM kernels; this file holds what surrounds them: im2col / col2im, the depth-wise 3x3 conv of
// Mlp (:85-133), attention against the 49 spatially-reduced keys (:148-208), bilinear resize (:581-608), the
// BatchNorm input gradient for the conv -> ReLU -> BN ordering (:31-46) and the sigmoid outputs (:635-639).
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

int grid_for(int64_t n, int cap = 8192) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------------------
// im2col: out[b,oy,ox, c*T + tap] (Kpad columns, zero padded), k = c*KH*KW + ky*KW + kx = OIHW flattening, so the
// convolution weight is the GEMM operand as it lies in memory.  Source: NCHW fp32 image or NHWC T activation.
// ------------------------------------------------------------------------------------------------
template <typename T, bool NCHW>
__global__ void im2col_kernel(const void* xin, T* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                              int pad, int Kpad) {
  const int taps = KH * KW, K = Cin * taps;
  const int64_t n = (int64_t)B * Ho * Wo * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = i % Kpad; int64_t r = i / Kpad;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    float v = 0.f;
    if (k < K) {
      const int c = k / taps, t = k - c * taps;
      const int ky = t / KW, kx = t - ky * KW;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
        if (NCHW) v = ((const float*)xin)[(((int64_t)b * Cin + c) * H + iy) * W + ix];
        else v = ElemTraits<T>::ld((const T*)xin + (((int64_t)b * H + iy) * W + ix) * Cin + c);
      }
    }
    ElemTraits<T>::st(out + i, v);
  }
}

// One 16-byte vector of consecutive k per thread (Kpad a multiple of the vector: every caller's): the index arithmetic of an output
// pixel -- five integer divisions by run-time values -- is paid once per 8 (bf16) elements instead of per element, the store is one
// 16-byte instruction; (c, ky, kx) advance by carries.  The ResNet stem's 7 x 7 stride-2 image (Unet / BIT-CD: 401 k pixels x 128
// columns) took 170 us per launch with the scalar kernel below, which stays for unaligned Kpad.
template <typename T, bool NCHW>
__global__ void im2col_vec_kernel(const void* xin, T* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                                  int pad, int Kpad) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int taps = KH * KW, K = Cin * taps, KV = Kpad / VEC;
  const int64_t n = (int64_t)B * Ho * Wo * KV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int kv = v % KV; int64_t r = v / KV;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    const int k0 = kv * VEC;
    int c = k0 / taps;
    const int t0 = k0 - c * taps;
    int ky = t0 / KW, kx = t0 - ky * KW;
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    float f[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float val = 0.f;
      const int iy = iy0 + ky, ix = ix0 + kx;
      if (k0 + j < K && iy >= 0 && iy < H && ix >= 0 && ix < W) {
        if (NCHW) val = ((const float*)xin)[(((int64_t)b * Cin + c) * H + iy) * W + ix];
        else val = ElemTraits<T>::ld((const T*)xin + (((int64_t)b * H + iy) * W + ix) * Cin + c);
      }
      f[j] = val;
      if (++kx == KW) { kx = 0; if (++ky == KH) { ky = 0; ++c; } }
    }
    *(u32x4*)(out + v * VEC) = vec_pack<T>(f);
  }
}

// col2im (adjoint): dx[b,iy,ix,c] = sum over taps with (iy + pad - ky) % stride == 0 of dcol[b,oy,ox,c*T+tap]
template <typename T>
__global__ void col2im_kernel(const T* dcol, T* dx, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                              int pad, int Kpad, int accumulate) {
  const int taps = KH * KW;
  const int64_t n = (int64_t)B * H * W * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % Cin; int64_t r = i / Cin;
    const int ix = r % W; r /= W;
    const int iy = r % H; const int b = r / H;
    float s = 0.f;
    for (int ky = 0; ky < KH; ++ky) {
      const int ty = iy + pad - ky;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int tx = ix + pad - kx;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= Wo) continue;
        s += ElemTraits<T>::ld(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * Kpad + c * taps + ky * KW + kx);
      }
    }
    if (accumulate) s += ElemTraits<T>::ld(dx + i);
    ElemTraits<T>::st(dx + i, s);
  }
}

// y = [relu](x * scale[c] + shift[c]) (scale/shift may be null = 1/0), times alpha: materialised BatchNorm output / scaled copy
template <typename T>
__global__ void affine_kernel(const T* x, const float* scale, const float* shift, T* y, int64_t nvec, int CV, int relu, float alpha) {
  constexpr int VEC = ElemTraits<T>::kVec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CV) * VEC;
    float f[VEC];
    vec_unpack<T>(*(const u32x4*)(x + v * VEC), f);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float o = f[j];
      if (scale) o = o * scale[c + j] + shift[c + j];
      if (relu) o = fmaxf(o, 0.f);
      f[j] = o * alpha;
    }
    *(u32x4*)(y + v * VEC) = vec_pack<T>(f);
  }
}

// ------------------------------------------------------------------------------------------------
// max pool 3x3 stride 2 pad 1 (ResNet stem).  Ho = (H + 1) / 2.  Backward (gather form): input pixel (iy, ix) collects dy of
// every window that contains it and whose FIRST maximum (scan order ky, kx) is this pixel.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool3s2_fwd_kernel(const T* x, T* y, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)B * Ho * Wo * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    float m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = -3.0e38f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float t[VEC];
        vec_unpack<T>(*(const u32x4*)(x + (((int64_t)b * H + iy) * W + ix) * C + cv * VEC), t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) m[j] = fmaxf(m[j], t[j]);
      }
    }
    *(u32x4*)(y + v * VEC) = vec_pack<T>(m);
  }
}
template <typename T>
__global__ void maxpool3s2_bwd_kernel(const T* x, const T* dy, T* dx, int accumulate, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)B * H * W * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ix = r % W; r /= W;
    const int iy = r % H; const int b = r / H;
    float me[VEC], s[VEC];
    vec_unpack<T>(*(const u32x4*)(x + v * VEC), me);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    for (int oy = (iy - 1 + 1) / 2; oy <= (iy + 1) / 2 && oy < Ho; ++oy) {          // windows with 2*oy-1 <= iy <= 2*oy+1
      if (2 * oy - 1 > iy) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2 && ox < Wo; ++ox) {
        if (2 * ox - 1 > ix) continue;
        // is (iy, ix) the first maximum of window (oy, ox)?  per channel: no earlier element >= me, no later element > me
        bool first[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) first[j] = true;
        for (int ky = 0; ky < 3; ++ky) {
          const int yy = 2 * oy - 1 + ky;
          if (yy < 0 || yy >= H) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int xx = 2 * ox - 1 + kx;
            if (xx < 0 || xx >= W || (yy == iy && xx == ix)) continue;
            const bool earlier = yy < iy || (yy == iy && xx < ix);
            float t[VEC];
            vec_unpack<T>(*(const u32x4*)(x + (((int64_t)b * H + yy) * W + xx) * C + cv * VEC), t);
#pragma unroll
            for (int j = 0; j < VEC; ++j) if (earlier ? t[j] >= me[j] : t[j] > me[j]) first[j] = false;
          }
        }
        float g[VEC];
        vec_unpack<T>(*(const u32x4*)(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + cv * VEC), g);
#pragma unroll
        for (int j = 0; j < VEC; ++j) if (first[j]) s[j] += g[j];
      }
    }
    if (accumulate) {
      float o[VEC];
      vec_unpack<T>(*(const u32x4*)(dx + v * VEC), o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += o[j];
    }
    *(u32x4*)(dx + v * VEC) = vec_pack<T>(s);
  }
}

// Training pair (round 5): the forward also records WHICH element of the window was the first maximum (one byte per output element,
// code = ky * 3 + kx), and the backward compares codes instead of re-deriving the first maximum from the eight other elements of up to
// four windows (the gather form above: <= 36 sixteen-byte loads per thread, 325 us at 112 x 112 x 64 x 32 images; this one: <= 4 x
// (8 + 16) bytes, the gradient of a window still goes to exactly the element torch picks -- ties are common behind a ReLU in bf16).
template <typename T>
__global__ void maxpool3s2_fwd_idx_kernel(const T* x, T* y, unsigned char* idx, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)B * Ho * Wo * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    float m[VEC];
    unsigned char code[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { m[j] = -3.0e38f; code[j] = 0; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float t[VEC];
        vec_unpack<T>(*(const u32x4*)(x + (((int64_t)b * H + iy) * W + ix) * C + cv * VEC), t);
#pragma unroll
        for (int j = 0; j < VEC; ++j) if (t[j] > m[j]) { m[j] = t[j]; code[j] = (unsigned char)(ky * 3 + kx); }   // strict: the FIRST maximum
      }
    }
    *(u32x4*)(y + v * VEC) = vec_pack<T>(m);
    unsigned char* ip = idx + v * VEC;
#pragma unroll
    for (int j = 0; j < VEC; j += 4)
      *(uint32_t*)(ip + j) = (uint32_t)code[j] | ((uint32_t)code[j + 1] << 8) | ((uint32_t)code[j + 2] << 16) | ((uint32_t)code[j + 3] << 24);
  }
}
template <typename T>
__global__ void maxpool3s2_bwd_idx_kernel(const unsigned char* idx, const T* dy, T* dx, int accumulate, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)B * H * W * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ix = r % W; r /= W;
    const int iy = r % H; const int b = r / H;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    // windows (oy, ox) with 2*oy - 1 <= iy <= 2*oy + 1: oy in {iy / 2, (iy + 1) / 2} (one window for odd... two for odd rows)
    const int oy0 = iy / 2, oy1 = (iy + 1) / 2, ox0 = ix / 2, ox1 = (ix + 1) / 2;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oy = a ? oy1 : oy0;
      if ((a && oy1 == oy0) || oy >= Ho) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int ox = c ? ox1 : ox0;
        if ((c && ox1 == ox0) || ox >= Wo) continue;
        const unsigned mycode = (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)));
        const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * C + cv * VEC;
        float g[VEC];
        vec_unpack<T>(*(const u32x4*)(dy + o), g);
        const unsigned char* ip = idx + o;
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
          const uint32_t w4 = *(const uint32_t*)(ip + j);
#pragma unroll
          for (int q = 0; q < 4; ++q) if (((w4 >> (8 * q)) & 0xffu) == mycode) s[j + q] += g[j + q];
        }
      }
    }
    if (accumulate) {
      float o[VEC];
      vec_unpack<T>(*(const u32x4*)(dx + v * VEC), o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += o[j];
    }
    *(u32x4*)(dx + v * VEC) = vec_pack<T>(s);
  }
}

// ------------------------------------------------------------------------------------------------
// depth-wise 3x3 conv, pad 1 (DWConv, changeformer.py:85-96): z = dw(x) + b ; g = gelu(z) (exact erf).
// MODE 0: forward (writes z and g) ; MODE 1: adjoint (flipped taps, no bias, writes z only)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// thread = (pixel lane tid / 32, channel vector blockIdx.y*32 + tid % 32): the 9 x VEC tap weights live in registers and the
// thread walks PPT pixels of its block's pixel slab (the per-pixel version re-fetched 72 weights per output vector)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* x, const float* w, const float* bias, T* z, T* g, int B, int H, int W, int C,
                                                        int pix_per_block) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int cv = blockIdx.y * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
  if (cv >= CV) return;
  const int c0 = cv * VEC;
  float wr[9][VEC], br[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    br[j] = (MODE == 0 && bias) ? bias[c0 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][j] = w[(c0 + j) * 9 + (MODE == 0 ? t : 8 - t)];
  }
  const int64_t npix = (int64_t)B * H * W;
  // consecutive pixel slabs share their halo rows: keep them on ONE XCD (workgroups go round-robin over the 8 XCDs, each with its own
  // L2; gridDim.x is a multiple of 8) -- measured 2.3x the algorithmic fetch without this mapping
  const int64_t slab = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int64_t p0 = slab * pix_per_block, p1 = min(npix, p0 + pix_per_block);
  for (int64_t p = p0 + pl; p < p1; p += 8) {
    const int px = p % W; const int64_t r = p / W;
    const int py = r % H;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = br[j];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dx = t % 3 - 1;
      const int iy = py + dy, ix = px + dx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      float xx[VEC];
      vec_unpack<T>(*(const u32x4*)(x + (p + (int64_t)dy * W + dx) * C + c0), xx);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += xx[j] * wr[t][j];
    }
    *(u32x4*)(z + p * C + c0) = vec_pack<T>(acc);
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = gelu_f(ElemTraits<T>::cvt(acc[j]));
      *(u32x4*)(g + p * C + c0) = vec_pack<T>(acc);
    }
  }
}

// Row-walking form of the two kernels around this comment (the default; KSMI_DW_ROW=0 selects the per-pixel forms).  A thread owns
// (channel vector, row segment of `seg` pixels) and slides a 3x3 window along the row: 3 new 16-byte loads per output instead of 9,
// the window and the 9 x VEC tap weights in registers, the loads of the next two columns in flight.  The per-pixel forms
// spend their time in the VALU (9 unpacks + 9 address computations per output vector): 1.6 TB/s on the stage-1 map of the MiT
// encoder (models/changeformer.py:85-96 DWConv inside Mlp :119-133).  Same tap order and fp32 operations as dwconv3x3_kernel
// (out-of-image taps contribute +0 instead of being skipped), so both forms give the same bits.
template <typename T>
struct DwCol {                                                  // one window column: three rows of one pixel, raw 16-byte vectors
  u32x4 r[3];
};
template <typename T>
__device__ __forceinline__ DwCol<T> dw_load_col(const T* const* rp, const bool* rv, int ix, int W, int C) {
  DwCol<T> c;
  const bool in = ix >= 0 && ix < W;
#pragma unroll
  for (int d = 0; d < 3; ++d) c.r[d] = (in && rv[d]) ? *(const u32x4*)(rp[d] + (int64_t)ix * C) : (u32x4){0u, 0u, 0u, 0u};
  return c;
}

// DROP (round 6): Mlp.drop behind the activation (models/changeformer.py:130) applied to g as it is stored -- element index = the flat
// index of the [tokens][C] matrix, the same draw ksmi_dropout_apply(g, site) makes in a pass of its own (13 launches and 2 x the
// 4C-wide tensor per ChangeFormer step); the product is formed on the ROUNDED activation like that pass did: bit-identical
template <typename T, int MODE, bool DROP = false>
__global__ __launch_bounds__(256) void dwconv3x3_row_kernel(const T* x, const float* w, const float* bias, T* z, T* g, int B, int H, int W, int C,
                                                            int seg, int nseg, int64_t units, uint32_t thr = 0, float inv = 1.f,
                                                            uint32_t site = 0, const uint32_t* __restrict__ rng = nullptr) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int cv = blockIdx.y * 32 + (threadIdx.x & 31), pl = threadIdx.x >> 5;
  if (cv >= CV) return;
  const int c0 = cv * VEC;
  // consecutive units are neighbouring rows of one image (shared halo rows): keep them on ONE XCD (gridDim.x is a multiple of 8)
  const int64_t u = ((int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 8 + pl;
  if (u >= units) return;
  const uint32_t dkey = DROP ? ksmi_rng_key(rng, site) : 0u;
  float wr[9][VEC], br[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    br[j] = (MODE == 0 && bias) ? bias[c0 + j] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][j] = w[(c0 + j) * 9 + (MODE == 0 ? t : 8 - t)];
  }
  const int sgi = (int)(u % nseg);
  const int64_t row = u / nseg;                                 // b * H + y
  const int y = (int)(row % H);
  const int x0 = sgi * seg, x1 = min(W, x0 + seg);
  const T* rp[3]; bool rv[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) { rv[d] = y + d - 1 >= 0 && y + d - 1 < H; rp[d] = x + (row + d - 1) * (int64_t)W * C + c0; }
  float win[3][3][VEC];                                         // [column slot][row][channel]
  {
    const DwCol<T> a = dw_load_col<T>(rp, rv, x0 - 1, W, C), b = dw_load_col<T>(rp, rv, x0, W, C);
#pragma unroll
    for (int d = 0; d < 3; ++d) { vec_unpack<T>(a.r[d], win[0][d]); vec_unpack<T>(b.r[d], win[1][d]); }
  }
  DwCol<T> nxt[2] = {dw_load_col<T>(rp, rv, x0 + 1, W, C), dw_load_col<T>(rp, rv, x0 + 2, W, C)};     // two columns in flight
  T* zo = z + (row * W) * (int64_t)C + c0;
  T* go = MODE == 0 ? g + (row * W) * (int64_t)C + c0 : nullptr;
  // slot roles rotate with the pixel: (left, mid, right) = (k, k+1, k+2) mod 3, prefetch slot k mod 2 -- unrolled by 6 so that
  // the indices are constants
#define KSMI_DW_STEP(L, M, R_, A)                                                                                \
  if (px < x1) {                                                                                                 \
    _Pragma("unroll") for (int d = 0; d < 3; ++d) vec_unpack<T>(nxt[A].r[d], win[R_][d]);                        \
    nxt[A] = dw_load_col<T>(rp, rv, px + 3, W, C);                                                               \
    float acc[VEC];                                                                                              \
    _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[j] = br[j];                                              \
    _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                              \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[j] += win[L][d][j] * wr[d * 3 + 0][j];                 \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[j] += win[M][d][j] * wr[d * 3 + 1][j];                 \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[j] += win[R_][d][j] * wr[d * 3 + 2][j];                \
    }                                                                                                            \
    *(u32x4*)(zo + (int64_t)px * C) = vec_pack<T>(acc);                                                          \
    if (MODE == 0) {                                                                                             \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[j] = gelu_f(ElemTraits<T>::cvt(acc[j]));               \
      if (DROP) {                                                                                                \
        const uint32_t e0 = (uint32_t)((row * W + px) * (int64_t)C + c0);                                        \
        _Pragma("unroll") for (int j = 0; j < VEC; ++j)                                                          \
          acc[j] = ksmi_rng_keep(dkey, e0 + j, thr) ? ElemTraits<T>::cvt(acc[j]) * inv : 0.f;                    \
      }                                                                                                          \
      *(u32x4*)(go + (int64_t)px * C) = vec_pack<T>(acc);                                                        \
    }                                                                                                            \
    ++px;                                                                                                        \
  }
  for (int px = x0; px < x1;) {
    KSMI_DW_STEP(0, 1, 2, 0)
    KSMI_DW_STEP(1, 2, 0, 1)
    KSMI_DW_STEP(2, 0, 1, 0)
    KSMI_DW_STEP(0, 1, 2, 1)
    KSMI_DW_STEP(1, 2, 0, 0)
    KSMI_DW_STEP(2, 0, 1, 1)
  }
#undef KSMI_DW_STEP
}

// partial[blockIdx.x][...] over the block's slab of row segments; thread = (segment lane t / CVB, channel vector t % CVB)
template <typename T, int CVB>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_row_kernel(const T* x, const T* dz, float* partial, int B, int H, int W, int C,
                                                                  int seg, int nseg, int64_t units) {
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int PL = 256 / CVB;
  __shared__ float red[256 * 5 * VEC];
  const int CV = C / VEC;
  const int cvl = threadIdx.x % CVB, cv = blockIdx.y * CVB + cvl, pl = threadIdx.x / CVB;
  const int64_t per = (units + gridDim.x - 1) / gridDim.x;
  const int64_t slab = (gridDim.x & 7) == 0 ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
  const int64_t u0 = per * slab, u1 = min(units, u0 + per);
  float acc[10][VEC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
  if (cv < CV) {
    const int c0 = cv * VEC;
    for (int64_t u = u0 + pl; u < u1; u += PL) {
      const int sgi = (int)(u % nseg);
      const int64_t row = u / nseg;
      const int y = (int)(row % H);
      const int x0 = sgi * seg, x1 = min(W, x0 + seg);
      const T* rp[3]; bool rv[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) { rv[d] = y + d - 1 >= 0 && y + d - 1 < H; rp[d] = x + (row + d - 1) * (int64_t)W * C + c0; }
      const T* dr = dz + (row * W) * (int64_t)C + c0;
      float win[3][3][VEC];
      {
        const DwCol<T> a = dw_load_col<T>(rp, rv, x0 - 1, W, C), b = dw_load_col<T>(rp, rv, x0, W, C);
#pragma unroll
        for (int d = 0; d < 3; ++d) { vec_unpack<T>(a.r[d], win[0][d]); vec_unpack<T>(b.r[d], win[1][d]); }
      }
      DwCol<T> nxt[2] = {dw_load_col<T>(rp, rv, x0 + 1, W, C), dw_load_col<T>(rp, rv, x0 + 2, W, C)};
      u32x4 dn[2];
      dn[0] = *(const u32x4*)(dr + (int64_t)x0 * C);
      dn[1] = x0 + 1 < x1 ? *(const u32x4*)(dr + (int64_t)(x0 + 1) * C) : (u32x4){0u, 0u, 0u, 0u};
#define KSMI_DW_STEP(L, M, R_, A)                                                                                \
  if (px < x1) {                                                                                                 \
    _Pragma("unroll") for (int d = 0; d < 3; ++d) vec_unpack<T>(nxt[A].r[d], win[R_][d]);                        \
    float dv[VEC];                                                                                               \
    vec_unpack<T>(dn[A], dv);                                                                                    \
    nxt[A] = dw_load_col<T>(rp, rv, px + 3, W, C);                                                               \
    if (px + 2 < x1) dn[A] = *(const u32x4*)(dr + (int64_t)(px + 2) * C);                                        \
    _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[9][j] += dv[j];                                          \
    _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                              \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[d * 3 + 0][j] += win[L][d][j] * dv[j];                 \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[d * 3 + 1][j] += win[M][d][j] * dv[j];                 \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[d * 3 + 2][j] += win[R_][d][j] * dv[j];                \
    }                                                                                                            \
    ++px;                                                                                                        \
  }
      for (int px = x0; px < x1;) {
        KSMI_DW_STEP(0, 1, 2, 0)
        KSMI_DW_STEP(1, 2, 0, 1)
        KSMI_DW_STEP(2, 0, 1, 0)
        KSMI_DW_STEP(0, 1, 2, 1)
        KSMI_DW_STEP(1, 2, 0, 0)
        KSMI_DW_STEP(2, 0, 1, 1)
      }
#undef KSMI_DW_STEP
    }
  }
  // block reduction over the PL segment lanes: two rounds of five accumulator rows through LDS, every thread sums some outputs
  float* prow = partial + (size_t)blockIdx.x * 10 * C;
  constexpr int RW = 5 * VEC;                                   // floats per thread and round
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    __syncthreads();
    float* mine = red + (pl * CVB + cvl) * RW;
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int j = 0; j < VEC; ++j) mine[t * VEC + j] = acc[r * 5 + t][j];
    __syncthreads();
    for (int o = threadIdx.x; o < CVB * RW; o += 256) {
      const int cl = o / RW, k = o - cl * RW, t = r * 5 + k / VEC, jj = k % VEC;
      const int cvo = blockIdx.y * CVB + cl;
      if (cvo >= CV) continue;
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < PL; ++q) sum += red[(q * CVB + cl) * RW + k];
      const int c = cvo * VEC + jj;
      if (t < 9) prow[c * 9 + t] = sum; else prow[9 * C + c] = sum;
    }
  }
}

// weight/bias gradient partials: partial[row][c*9 + t] = sum_p x[p + off_t][c] * dz[p][c] ; partial[row][9C + c] = sum_p dz[p][c]
// grid (rows, ceil(CV/CVB)); thread = (pixel lane t / CVB, channel vector t % CVB); CVB = 32 for <= 32 channel vectors (stage 1 of the
// encoder, the largest map: all 256 threads busy instead of half of them), else 64
template <typename T, int CVB>
__global__ void dwconv3x3_wgrad_kernel(const T* x, const T* dz, float* partial, int B, int H, int W, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int PL = 256 / CVB;
  __shared__ float red[PL][CVB][VEC + 1];
  const int CV = C / VEC;
  const int cvl = threadIdx.x % CVB, cv = blockIdx.y * CVB + cvl, pl = threadIdx.x / CVB;
  const int64_t npix = (int64_t)B * H * W;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
  // (XCD-aware slab order as in dwconv3x3_kernel when the row count allows it; the partial row index stays blockIdx.x)
  const int64_t slab = (gridDim.x & 7) == 0 ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
  const int64_t p0 = per * slab, p1 = min(npix, p0 + per);
  float acc[10][VEC];
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[t][j] = 0.f;
  if (cv < CV) {
    for (int64_t p = p0 + pl; p < p1; p += PL) {
      const int px = p % W; const int64_t r = p / W;
      const int py = r % H;
      float d[VEC];
      vec_unpack<T>(*(const u32x4*)(dz + p * C + cv * VEC), d);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[9][j] += d[j];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        float xx[VEC];
        vec_unpack<T>(*(const u32x4*)(x + (p + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * C + cv * VEC), xx);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[t][j] += xx[j] * d[j];
      }
    }
  }
  float* prow = partial + (size_t)blockIdx.x * 10 * C;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[pl][cvl][j] = acc[t][j];
    __syncthreads();
    if (pl == 0 && cv < CV) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < PL; ++k) s += red[k][cvl][j];
        const int c = cv * VEC + j;
        if (t < 9) prow[c * 9 + t] = s; else prow[9 * C + c] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Attention against NK spatially-reduced keys (changeformer.py:190-207): q [B*Nq][C], kv [B*NK][2C] in "(2 h d)" order,
// out [B*Nq][C]; head h uses columns h*D..h*D+D.  One query row per thread, K and V of (b, h) live in LDS as fp32.
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int NK>
__device__ __forceinline__ void load_kv(const T* kv, float* sK, float* sV, int b, int h, int C) {
  for (int i = threadIdx.x; i < NK * D; i += blockDim.x) {
    const int j = i / D, d = i - j * D;
    const T* row = kv + ((int64_t)b * NK + j) * 2 * C + h * D + d;
    sK[i] = ElemTraits<T>::ld(row);
    sV[i] = ElemTraits<T>::ld(row + C);
  }
}

// dot product of a register vector with a row of an LDS matrix (16-byte broadcast reads)
template <int D>
__device__ __forceinline__ float dot_row(const float* qr, const float* row) {
  float a = 0.f;
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const f32x4 k4 = *(const f32x4*)(row + d);
    a += qr[d] * k4[0] + qr[d + 1] * k4[1] + qr[d + 2] * k4[2] + qr[d + 3] * k4[3];
  }
  return a;
}
template <int D>
__device__ __forceinline__ void axpy_row(float* o, float p, const float* row) {
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const f32x4 v4 = *(const f32x4*)(row + d);
    o[d] += p * v4[0]; o[d + 1] += p * v4[1]; o[d + 2] += p * v4[2]; o[d + 3] += p * v4[3];
  }
}

struct SrDrop { uint32_t thr, site; float inv; const uint32_t* rng; };   // attn_drop (:160,203): thr = 0 -> off

// forward: one query per thread, keys walked with an online softmax (no per-key register arrays: the fully unrolled
// 49 x D version spilled kilobytes of scratch per thread)
template <typename T, int D, int NK>
__global__ __launch_bounds__(256) void sr_attn_fwd_kernel(const T* q, const T* kv, T* out, int Nq, int C, float scale, SrDrop dr) {
  __shared__ __attribute__((aligned(16))) float sK[NK * D];
  __shared__ __attribute__((aligned(16))) float sV[NK * D];
  const int b = blockIdx.z, h = blockIdx.y;
  load_kv<T, D, NK>(kv, sK, sV, b, h, C);
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nq) return;
  const T* qp = q + ((int64_t)b * Nq + i) * C + h * D;
  float qr[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { qr[d] = ElemTraits<T>::ld(qp + d) * scale; o[d] = 0.f; }
  float m = -3.0e38f, l = 0.f;
  const uint32_t dkey = dr.thr ? ksmi_rng_key(dr.rng, dr.site) : 0u;
  const uint32_t dbase = (((uint32_t)b * gridDim.y + h) * Nq + i) * NK;
#pragma unroll 1
  for (int j = 0; j < NK; ++j) {
    const float a = dot_row<D>(qr, sK + j * D);
    if (a > m) {
      const float c = __expf(m - a);
      l *= c;
#pragma unroll
      for (int d = 0; d < D; ++d) o[d] *= c;
      m = a;
    }
    const float p = __expf(a - m);
    l += p;                                                              // attn_drop acts on the normalised row: l stays whole
    const float pd = !dr.thr ? p : (ksmi_rng_keep(dkey, dbase + j, dr.thr) ? p * dr.inv : 0.f);
    axpy_row<D>(o, pd, sV + j * D);
  }
  const float inv = 1.f / l;
  T* op = out + ((int64_t)b * Nq + i) * C + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) ElemTraits<T>::st(op + d, o[d] * inv);
}

// backward, pass A (one query per thread): dq ; P and dS (scale folded in) to scratch [B][H][Nq][NK] fp32.
// Three walks over the keys: (1) softmax max / sum, (2) P, dP (parked in the scratch rows) and delta, (3) dS and dq.
template <typename T, int D, int NK>
__global__ __launch_bounds__(256) void sr_attn_bwd_q_kernel(const T* q, const T* kv, const T* dout, T* dq, float* Pbuf, float* dSbuf,
                                                            int Nq, int C, int H, float scale, SrDrop dr) {
  __shared__ __attribute__((aligned(16))) float sK[NK * D];
  __shared__ __attribute__((aligned(16))) float sV[NK * D];
  const int b = blockIdx.z, h = blockIdx.y;
  load_kv<T, D, NK>(kv, sK, sV, b, h, C);
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Nq) return;
  const T* qp = q + ((int64_t)b * Nq + i) * C + h * D;
  const T* gp = dout + ((int64_t)b * Nq + i) * C + h * D;
  float* Pp = Pbuf + (((int64_t)b * H + h) * Nq + i) * NK;
  float* Sp = dSbuf + (((int64_t)b * H + h) * Nq + i) * NK;
  float qr[D], gr[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { qr[d] = ElemTraits<T>::ld(qp + d) * scale; gr[d] = ElemTraits<T>::ld(gp + d); }
  float m = -3.0e38f;
  const uint32_t dkey = dr.thr ? ksmi_rng_key(dr.rng, dr.site) : 0u;
  const uint32_t dbase = (((uint32_t)b * H + h) * Nq + i) * NK;
#pragma unroll 1
  for (int j = 0; j < NK; ++j) {
    const float a = dot_row<D>(qr, sK + j * D);
    Pp[j] = a;
    m = fmaxf(m, a);
  }
  float l = 0.f;
#pragma unroll 1
  for (int j = 0; j < NK; ++j) l += __expf(Pp[j] - m);
  const float inv = 1.f / l;
  float delta = 0.f;
#pragma unroll 1
  for (int j = 0; j < NK; ++j) {
    const float p = __expf(Pp[j] - m) * inv;
    float dp = dot_row<D>(gr, sV + j * D);
    if (dr.thr) dp = ksmi_rng_keep(dkey, dbase + j, dr.thr) ? dp * dr.inv : 0.f;
    Pp[j] = p;
    Sp[j] = dp;
    delta += p * dp;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) gr[d] = 0.f;                              // gr now accumulates dq
#pragma unroll 1
  for (int j = 0; j < NK; ++j) {
    const float ds = Pp[j] * (Sp[j] - delta) * scale;
    Sp[j] = ds;
    if (dr.thr) Pp[j] = ksmi_rng_keep(dkey, dbase + j, dr.thr) ? Pp[j] * dr.inv : 0.f;   // the kv pass (dV) reads the dropped row
    axpy_row<D>(gr, ds, sK + j * D);
  }
  T* op = dq + ((int64_t)b * Nq + i) * C + h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) ElemTraits<T>::st(op + d, gr[d]);
}

// backward, pass B: partial[split][b][j][2C] : dK[j, h*D+d] = sum_i dS[i][j] q[i][d] ; dV[j, C + h*D+d] = sum_i P[i][j] dO[i][d]
// grid (nsplit, H, B); 256 threads; queries staged 32 at a time in LDS
template <typename T, int D, int NK>
__global__ __launch_bounds__(256) void sr_attn_bwd_kv_kernel(const T* q, const T* dout, const float* Pbuf, const float* dSbuf,
                                                             float* partial, int Nq, int C, int H, int B) {
  constexpr int QT = 32;
  __shared__ float sQ[QT * D], sG[QT * D], sP[QT * NK], sS[QT * NK];
  const int b = blockIdx.z, h = blockIdx.y, split = blockIdx.x, nsplit = gridDim.x;
  const int per = (Nq + nsplit - 1) / nsplit;
  const int i0 = split * per, i1 = min(Nq, i0 + per);
  constexpr int OUT = NK * D;                       // outputs per matrix
  constexpr int PER_T = (OUT + 255) / 256;
  float aK[PER_T], aV[PER_T];
#pragma unroll
  for (int k = 0; k < PER_T; ++k) { aK[k] = 0.f; aV[k] = 0.f; }
  for (int base = i0; base < i1; base += QT) {
    const int nq = min(QT, i1 - base);
    __syncthreads();
    for (int e = threadIdx.x; e < QT * D; e += 256) {
      const int qi = e / D, d = e - qi * D;
      const bool ok = qi < nq;
      const int64_t row = ((int64_t)b * Nq + base + qi) * C + h * D + d;
      sQ[e] = ok ? ElemTraits<T>::ld(q + row) : 0.f;
      sG[e] = ok ? ElemTraits<T>::ld(dout + row) : 0.f;
    }
    for (int e = threadIdx.x; e < QT * NK; e += 256) {
      const int qi = e / NK, j = e - qi * NK;
      const bool ok = qi < nq;
      const int64_t idx = (((int64_t)b * H + h) * Nq + base + qi) * NK + j;
      sP[e] = ok ? Pbuf[idx] : 0.f;
      sS[e] = ok ? dSbuf[idx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER_T; ++k) {
      const int o = threadIdx.x + k * 256;
      if (o < OUT) {
        const int j = o / D, d = o - j * D;
        float ak = 0.f, av = 0.f;
#pragma unroll 8
        for (int qi = 0; qi < QT; ++qi) {
          ak += sS[qi * NK + j] * sQ[qi * D + d];
          av += sP[qi * NK + j] * sG[qi * D + d];
        }
        aK[k] += ak; aV[k] += av;
      }
    }
  }
  float* pr = partial + ((size_t)split * B + b) * NK * 2 * C;
#pragma unroll
  for (int k = 0; k < PER_T; ++k) {
    const int o = threadIdx.x + k * 256;
    if (o < OUT) {
      const int j = o / D, d = o - j * D;
      pr[(size_t)j * 2 * C + h * D + d] = aK[k];
      pr[(size_t)j * 2 * C + C + h * D + d] = aV[k];
    }
  }
}

// dkv[r][c] = sum_split partial[split][r][c]  (r over B*NK rows, c over 2C)
template <typename T>
__global__ void sum_splits_kernel(const float* partial, T* out, int64_t n, int nsplit) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * n + i];
    ElemTraits<T>::st(out + i, s);
  }
}

// ------------------------------------------------------------------------------------------------
// bilinear resize, align_corners=False (F.interpolate semantics: src = max(0, (o+0.5)*in/out - 0.5))
// forward: y = [add +] resize(x) ; backward (adjoint, gather form): dx (+)= resize^T(dy)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bil_src(int o, float ratio, int in, int& i0, int& i1, float& l) {
  float s = ((float)o + 0.5f) * ratio - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l = s - (float)i0;
}

template <typename T>
__global__ void bilinear_fwd_kernel(const T* x, const T* add, T* y, int B, int Hi, int Wi, int Ho, int Wo, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  const int64_t n = (int64_t)B * Ho * Wo * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ox = r % Wo; r /= Wo;
    const int oy = r % Ho; const int b = r / Ho;
    int y0, y1, x0, x1; float ly, lx;
    bil_src(oy, ry, Hi, y0, y1, ly);
    bil_src(ox, rx, Wi, x0, x1, lx);
    float a[VEC], bb[VEC], c[VEC], d[VEC], o[VEC];
    const T* base = x + (int64_t)b * Hi * Wi * C + cv * VEC;
    vec_unpack<T>(*(const u32x4*)(base + ((int64_t)y0 * Wi + x0) * C), a);
    vec_unpack<T>(*(const u32x4*)(base + ((int64_t)y0 * Wi + x1) * C), bb);
    vec_unpack<T>(*(const u32x4*)(base + ((int64_t)y1 * Wi + x0) * C), c);
    vec_unpack<T>(*(const u32x4*)(base + ((int64_t)y1 * Wi + x1) * C), d);
    if (add) vec_unpack<T>(*(const u32x4*)(add + v * VEC), o);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float top = a[j] + lx * (bb[j] - a[j]), bot = c[j] + lx * (d[j] - c[j]);
      const float val = top + ly * (bot - top);
      o[j] = add ? o[j] + val : val;
    }
    *(u32x4*)(y + v * VEC) = vec_pack<T>(o);
  }
}

template <typename T>
__global__ void bilinear_bwd_kernel(const T* dy, T* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int accumulate) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const float ry = (float)Hi / (float)Ho, rx = (float)Wi / (float)Wo;
  const int64_t n = (int64_t)B * Hi * Wi * CV;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
    const int cv = v % CV; int64_t r = v / CV;
    const int ix = r % Wi; r /= Wi;
    const int iy = r % Hi; const int b = r / Hi;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    // output rows whose two taps can include iy: src = (o + 0.5) * r - 0.5 in (iy - 1, iy + 1)  ->  o in ((iy - 0.5) / r - 0.5,
    // (iy + 1.5) / r - 0.5), one more on each side for the rounding of r (2 s + 2 candidates per axis; the first version walked 5 s:
    // 1600 candidates per input pixel on the 7 -> 56 map, 190 us)
    const int oy_lo = max(0, (int)floorf(((float)iy - 0.5f) / ry - 0.5f) - 1), oy_hi = min(Ho - 1, (int)ceilf(((float)iy + 1.5f) / ry - 0.5f) + 1);
    const int ox_lo = max(0, (int)floorf(((float)ix - 0.5f) / rx - 0.5f) - 1), ox_hi = min(Wo - 1, (int)ceilf(((float)ix + 1.5f) / rx - 0.5f) + 1);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      bil_src(oy, ry, Hi, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        bil_src(ox, rx, Wi, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx == 0.f) continue;
        float g[VEC];
        vec_unpack<T>(*(const u32x4*)(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + cv * VEC), g);
        const float w = wy * wx;
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += w * g[j];
      }
    }
    if (accumulate) {
      float o[VEC];
      vec_unpack<T>(*(const u32x4*)(dx + v * VEC), o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += o[j];
    }
    *(u32x4*)(dx + v * VEC) = vec_pack<T>(s);
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm input gradient for y = BN(r), r = relu(v) (or r = v): dv = gamma*rstd*(dy - s0/n - rhat*s1/n) [* (r > 0)]
// sums[0][c] = sum dy, sums[1][c] = sum dy*rhat (from the consumer's dgrad epilogue + ksmi_reduce_rows)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* dy, const T* r, const float* mean, const float* rstd, const float* gamma,
                                    const float* sums, T* dv, int relu_mask, float inv_n, int64_t nvec, int CV, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(v % CV) * VEC;
    float g[VEC], rr[VEC];
    vec_unpack<T>(*(const u32x4*)(dy + v * VEC), g);
    vec_unpack<T>(*(const u32x4*)(r + v * VEC), rr);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float rh = (rr[j] - mean[c + j]) * rstd[c + j];
      float o = gamma[c + j] * rstd[c + j] * (g[j] - sums[c + j] * inv_n - rh * sums[C + c + j] * inv_n);
      if (relu_mask && !(rr[j] > 0.f)) o = 0.f;
      g[j] = o;
    }
    *(u32x4*)(dv + v * VEC) = vec_pack<T>(g);
  }
}

// partial[row][0][c] = sum dy, partial[row][1][c] = sum dy * xhat over the row's pixel slab (plain BatchNorm, no ReLU mask):
// grid (rows, ceil(CV/64)); thread = (pixel lane t>>6, channel vector t&63)
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* dy, const T* x, const float* mean, const float* rstd, float* partial, int64_t npix, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  __shared__ float red[4][64][2 * VEC + 1];
  const int CV = C / VEC;
  const int cvl = threadIdx.x & 63, cv = blockIdx.y * 64 + cvl, pl = threadIdx.x >> 6;
  float a[VEC], b[VEC], mu[VEC], rs[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { a[j] = 0.f; b[j] = 0.f; mu[j] = cv < CV ? mean[cv * VEC + j] : 0.f; rs[j] = cv < CV ? rstd[cv * VEC + j] : 0.f; }
  if (cv < CV) {
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = per * blockIdx.x, p1 = min(npix, p0 + per);
    for (int64_t p = p0 + pl; p < p1; p += 4) {
      float g[VEC], xv[VEC];
      vec_unpack<T>(*(const u32x4*)(dy + p * C + cv * VEC), g);
      vec_unpack<T>(*(const u32x4*)(x + p * C + cv * VEC), xv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { a[j] += g[j]; b[j] += g[j] * (xv[j] - mu[j]) * rs[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { red[pl][cvl][j] = a[j]; red[pl][cvl][VEC + j] = b[j]; }
  __syncthreads();
  if (pl == 0 && cv < CV) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      partial[((size_t)blockIdx.x * 2 + 0) * C + cv * VEC + j] = red[0][cvl][j] + red[1][cvl][j] + red[2][cvl][j] + red[3][cvl][j];
      partial[((size_t)blockIdx.x * 2 + 1) * C + cv * VEC + j] = red[0][cvl][VEC + j] + red[1][cvl][VEC + j] + red[2][cvl][VEC + j] + red[3][cvl][VEC + j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SAR tile preprocessing of the reference's Dataset (dataset/Dataset.py:164-168,193-198): clamp to [0, clamp_input],
// NaN -> clamp_input (torch.nan_to_num(image, clamp_input): +inf/-inf were already clamped), (x - mean[c]) / std[c]; NCHW fp32
// ------------------------------------------------------------------------------------------------
__global__ void sar_preprocess_kernel(const float* x, const float* mean, const float* stdv, float* y, int C, int64_t HW, int64_t n, float clampv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % C);
    float v = x[i];
    if (clampv >= 0.f) v = (v != v) ? clampv : fminf(fmaxf(v, 0.f), clampv);
    y[i] = (v - mean[c]) / stdv[c];
  }
}

// ------------------------------------------------------------------------------------------------
// head output NHWC [B][HW][Cs] -> NCHW fp32 with optional sigmoid (changeformer.py:635-639) and its adjoint
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void out_to_nchw_kernel(const T* x, float* y, int B, int C, int Cs, int64_t HW, int act) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % HW; int64_t r = i / HW;
    const int c = r % C; const int b = r / C;
    const T* xp = x + ((int64_t)b * HW + p) * Cs;
    float v = ElemTraits<T>::ld(xp + c);
    if (act == 1) v = 1.f / (1.f + __expf(-v));
    else if (act == 2 || act == 3) {           // nn.Softmax(dim=1) (siam_conc.py:93,177) / nn.LogSoftmax(dim=1) (siam_diff.py:93,173)
      float m = v;
      for (int k = 0; k < C; ++k) m = fmaxf(m, ElemTraits<T>::ld(xp + k));
      float l = 0.f;
      for (int k = 0; k < C; ++k) l += __expf(ElemTraits<T>::ld(xp + k) - m);
      v = act == 2 ? __expf(v - m) / l : v - m - __logf(l);
    }
    y[i] = v;
  }
}
template <typename T>
__global__ void dout_to_nhwc_kernel(const float* dy, const float* y, T* dx, int B, int C, int Cs, int64_t HW, int act) {
  const int64_t n = (int64_t)B * HW * Cs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = i % Cs; int64_t r = i / Cs;
    const int64_t p = r % HW; const int b = r / HW;
    float v = 0.f;
    if (c < C) {
      const int64_t j = ((int64_t)b * C + c) * HW + p;
      v = dy[j];
      if (act == 1) { const float s = y[j]; v *= s * (1.f - s); }
      else if (act == 2) {                     // softmax adjoint: y_c (dy_c - sum_k y_k dy_k)
        float dot = 0.f;
        for (int k = 0; k < C; ++k) { const int64_t jk = ((int64_t)b * C + k) * HW + p; dot += y[jk] * dy[jk]; }
        v = y[j] * (v - dot);
      } else if (act == 3) {                   // log-softmax adjoint: dy_c - exp(y_c) sum_k dy_k
        float tot = 0.f;
        for (int k = 0; k < C; ++k) tot += dy[((int64_t)b * C + k) * HW + p];
        v = v - __expf(y[j]) * tot;
      }
    }
    ElemTraits<T>::st(dx + i, v);
  }
}

template <typename T, int D>
int sr_attn_launch(int which, const void* q, const void* kv, const void* io, void* o2, float* Pb, float* Sb, float* partial,
                   int B, int Nq, int H, int C, float scale, int nsplit, SrDrop dr, hipStream_t st) {
  constexpr int NK = 49;
  const dim3 grid((Nq + 255) / 256, H, B);
  if (which == 0) hipLaunchKernelGGL((sr_attn_fwd_kernel<T, D, NK>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (T*)o2, Nq, C, scale, dr);
  else if (which == 1)
    hipLaunchKernelGGL((sr_attn_bwd_q_kernel<T, D, NK>), grid, dim3(256), 0, st, (const T*)q, (const T*)kv, (const T*)io, (T*)o2, Pb, Sb, Nq, C, H, scale, dr);
  else
    hipLaunchKernelGGL((sr_attn_bwd_kv_kernel<T, D, NK>), dim3(nsplit, H, B), dim3(256), 0, st, (const T*)q, (const T*)io, Pb, Sb, partial, Nq, C, H, B);
  return ksmi_check_launch("sr_attention");
}

int sr_attn_dispatch(int which, const void* q, const void* kv, const void* io, void* o2, float* Pb, float* Sb, float* partial,
                     int B, int Nq, int Nk, int H, int C, float scale, int nsplit, int dtype, SrDrop dr, hipStream_t st) {
  if (Nk != 49) return ksmi_fail(KSMI_E_UNSUPPORTED, "sr_attention: specialised for 49 keys (224x224 tiles: 7x7 after spatial reduction)");
  if (H < 1 || C % H) return ksmi_fail(KSMI_E_ARG, "sr_attention: C must be divisible by heads");
  const int D = C / H;
  if (dtype != KSMI_BF16 && dtype != KSMI_F32) return ksmi_fail(KSMI_E_ARG, "bad dtype");
  if (D == 64) return dtype == KSMI_BF16 ? sr_attn_launch<bf16_t, 64>(which, q, kv, io, o2, Pb, Sb, partial, B, Nq, H, C, scale, nsplit, dr, st)
                                         : sr_attn_launch<float, 64>(which, q, kv, io, o2, Pb, Sb, partial, B, Nq, H, C, scale, nsplit, dr, st);
  if (D == 80) return dtype == KSMI_BF16 ? sr_attn_launch<bf16_t, 80>(which, q, kv, io, o2, Pb, Sb, partial, B, Nq, H, C, scale, nsplit, dr, st)
                                         : sr_attn_launch<float, 80>(which, q, kv, io, o2, Pb, Sb, partial, B, Nq, H, C, scale, nsplit, dr, st);
  return ksmi_fail(KSMI_E_UNSUPPORTED, "sr_attention: head dim must be 64 or 80 (ChangeFormerV6 embed_dims / num_heads)");
}

}  // namespace

#define KSMI_DT(dtype, EXPR_BF16, EXPR_F32)                                  \
  do {                                                                       \
    if ((dtype) == KSMI_BF16) { EXPR_BF16; }                                 \
    else if ((dtype) == KSMI_F32) { EXPR_F32; }                              \
    else return ksmi_fail(KSMI_E_ARG, "bad dtype");                          \
  } while (0)

// ---- channel-fastest im2col family (K index = tap * Cin + c) ---------------------------------------------------------
// The OIHW-flattened order above (k = c * taps + tap) makes both directions 2-byte gathers at a stride of `taps` elements
// (col2im: 181 us per ChangeFormer patch embedding).  With the channel fastest every (pixel, tap) pair moves whole 16-byte
// channel vectors, coalesced on both sides; the GEMM weight is re-ordered to match (weight_to_tc, a few MB per step) and the
// weight gradient is put back in OIHW order by grad_from_tc.
template <typename T>
__global__ void im2col_tc_kernel(const T* x, T* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                                 int Kpad) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int KV = Kpad / VEC, CV = Cin / VEC, taps = KH * KW;
  const int64_t n = (int64_t)B * Ho * Wo * KV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kv = (int)(i % KV); int64_t r = i / KV;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int b = (int)(r / Ho);
    u32x4 v = (u32x4){0u, 0u, 0u, 0u};
    const int t = kv / CV, cv = kv - t * CV;
    if (t < taps) {
      const int ky = t / KW, kx = t - ky * KW;
      const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const u32x4*)(x + (((int64_t)b * H + iy) * W + ix) * Cin + cv * VEC);
    }
    *(u32x4*)(out + i * VEC) = v;
  }
}

template <typename T>
__global__ void col2im_tc_kernel(const T* dcol, T* dx, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                                 int Kpad, int accumulate) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = Cin / VEC;
  const int64_t n = (int64_t)B * H * W * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(i % CV); int64_t r = i / CV;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H); const int b = (int)(r / H);
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    for (int ky = 0; ky < KH; ++ky) {
      const int ty = iy + pad - ky;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < KW; ++kx) {
        const int tx = ix + pad - kx;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= Wo) continue;
        float f[VEC];
        vec_unpack<T>(*(const u32x4*)(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * Kpad + (ky * KW + kx) * Cin + cv * VEC), f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += f[j];
      }
    }
    if (accumulate) {
      float o[VEC];
      vec_unpack<T>(*(const u32x4*)(dx + i * VEC), o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += o[j];
    }
    *(u32x4*)(dx + i * VEC) = vec_pack<T>(s);
  }
}

// out[n][t * Cin + c] = w[n][c][t] (activation dtype, zero in the K padding); grad[n][c][t] (+)= g[n][t * Cin + c]
template <typename T>
__global__ void weight_to_tc_kernel(const float* w, T* out, int N, int Cin, int taps, int Kpad) {
  const int64_t n = (int64_t)N * Kpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad); const int o = (int)(i / Kpad);
    const int t = k / Cin, c = k - t * Cin;
    ElemTraits<T>::st(out + i, t < taps ? w[((int64_t)o * Cin + c) * taps + t] : 0.f);
  }
}
__global__ void grad_from_tc_kernel(const float* g, float* grad, int N, int Cin, int taps, int Kpad, int accumulate) {
  const int64_t n = (int64_t)N * Cin * taps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % taps); int64_t r = i / taps;
    const int c = (int)(r % Cin); const int o = (int)(r / Cin);
    const float v = g[(int64_t)o * Kpad + t * Cin + c];
    grad[i] = accumulate ? grad[i] + v : v;
  }
}

extern "C" {

int ksmi_im2col_tc(const void* x, void* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int Kpad,
                   int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (Cin % vec || Kpad % vec || Kpad < Cin * KH * KW) return ksmi_fail(KSMI_E_ARG, "im2col_tc: Cin and Kpad must be multiples of the 16-byte vector");
  const int64_t n = (int64_t)B * Ho * Wo * (Kpad / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(im2col_tc_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad),
          hipLaunchKernelGGL(im2col_tc_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)x, (float*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad));
  return ksmi_check_launch("im2col_tc");
}

int ksmi_col2im_tc(const void* dcol, void* dx, int accumulate, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride,
                   int pad, int Kpad, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (Cin % vec || Kpad % vec || Kpad < Cin * KH * KW) return ksmi_fail(KSMI_E_ARG, "col2im_tc: Cin and Kpad must be multiples of the 16-byte vector");
  const int64_t n = (int64_t)B * H * W * (Cin / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(col2im_tc_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)dcol, (bf16_t*)dx, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad, accumulate),
          hipLaunchKernelGGL(col2im_tc_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)dcol, (float*)dx, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad, accumulate));
  return ksmi_check_launch("col2im_tc");
}

int ksmi_weight_to_tc(const float* w, void* out, int N, int Cin, int taps, int Kpad, int dtype, void* stream) {
  if (Kpad < Cin * taps) return ksmi_fail(KSMI_E_ARG, "weight_to_tc: Kpad < Cin*taps");
  const int64_t n = (int64_t)N * Kpad;
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(weight_to_tc_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, w, (bf16_t*)out, N, Cin, taps, Kpad),
          hipLaunchKernelGGL(weight_to_tc_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, w, (float*)out, N, Cin, taps, Kpad));
  return ksmi_check_launch("weight_to_tc");
}

int ksmi_grad_from_tc(const float* g, float* grad, int N, int Cin, int taps, int Kpad, int accumulate, void* stream) {
  if (Kpad < Cin * taps) return ksmi_fail(KSMI_E_ARG, "grad_from_tc: Kpad < Cin*taps");
  const int64_t n = (int64_t)N * Cin * taps;
  hipLaunchKernelGGL(grad_from_tc_kernel, dim3(grid_for(n, 65536)), dim3(256), 0, (hipStream_t)stream, g, grad, N, Cin, taps, Kpad, accumulate);
  return ksmi_check_launch("grad_from_tc");
}

int ksmi_im2col(const void* x, void* out, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                int Kpad, int src_nchw_f32, int dtype, void* stream) {
  if (Kpad < Cin * KH * KW) return ksmi_fail(KSMI_E_ARG, "im2col: Kpad < Cin*KH*KW");
  const int64_t n = (int64_t)B * Ho * Wo * Kpad;
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  static const bool scalar_only = ksmi_knob_is_set("KSMI_IM2COL_SCALAR");      // A/B switch
  if (Kpad % vec == 0 && !scalar_only && ((uintptr_t)out & 15) == 0) {
    const int64_t nv = n / vec;
    if (src_nchw_f32) {
      KSMI_DT(dtype,
              hipLaunchKernelGGL((im2col_vec_kernel<bf16_t, true>), dim3(grid_for(nv, 65536)), dim3(256), 0, st, x, (bf16_t*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad),
              hipLaunchKernelGGL((im2col_vec_kernel<float, true>), dim3(grid_for(nv, 65536)), dim3(256), 0, st, x, (float*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad));
    } else {
      KSMI_DT(dtype,
              hipLaunchKernelGGL((im2col_vec_kernel<bf16_t, false>), dim3(grid_for(nv, 65536)), dim3(256), 0, st, x, (bf16_t*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad),
              hipLaunchKernelGGL((im2col_vec_kernel<float, false>), dim3(grid_for(nv, 65536)), dim3(256), 0, st, x, (float*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad));
    }
    return ksmi_check_launch("im2col");
  }
  if (src_nchw_f32) {
    KSMI_DT(dtype,
            hipLaunchKernelGGL((im2col_kernel<bf16_t, true>), dim3(grid_for(n, 65536)), dim3(256), 0, st, x, (bf16_t*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad),
            hipLaunchKernelGGL((im2col_kernel<float, true>), dim3(grid_for(n, 65536)), dim3(256), 0, st, x, (float*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad));
  } else {
    KSMI_DT(dtype,
            hipLaunchKernelGGL((im2col_kernel<bf16_t, false>), dim3(grid_for(n, 65536)), dim3(256), 0, st, x, (bf16_t*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad),
            hipLaunchKernelGGL((im2col_kernel<float, false>), dim3(grid_for(n, 65536)), dim3(256), 0, st, x, (float*)out, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad));
  }
  return ksmi_check_launch("im2col");
}

int ksmi_maxpool3x3s2_forward(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "maxpool3x3s2: C must be a multiple of the 16-byte vector");
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(maxpool3s2_fwd_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C),
          hipLaunchKernelGGL(maxpool3s2_fwd_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C));
  return ksmi_check_launch("maxpool3x3s2_fwd");
}

int ksmi_maxpool3x3s2_backward(const void* x, const void* dy, void* dx, int accumulate, int B, int H, int W, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "maxpool3x3s2: C must be a multiple of the 16-byte vector");
  const int64_t n = (int64_t)B * H * W * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(maxpool3s2_bwd_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, accumulate, B, H, W, C),
          hipLaunchKernelGGL(maxpool3s2_bwd_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, accumulate, B, H, W, C));
  return ksmi_check_launch("maxpool3x3s2_bwd");
}

int ksmi_maxpool3x3s2_forward_idx(const void* x, void* y, void* idx, int B, int H, int W, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (!x || !y || !idx || C % vec) return ksmi_fail(KSMI_E_ARG, "maxpool3x3s2_idx: C must be a multiple of the 16-byte vector");
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(maxpool3s2_fwd_idx_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)idx, B, H, W, C),
          hipLaunchKernelGGL(maxpool3s2_fwd_idx_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)x, (float*)y, (unsigned char*)idx, B, H, W, C));
  return ksmi_check_launch("maxpool3x3s2_fwd_idx");
}

int ksmi_maxpool3x3s2_backward_idx(const void* idx, const void* dy, void* dx, int accumulate, int B, int H, int W, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (!idx || !dy || !dx || C % vec) return ksmi_fail(KSMI_E_ARG, "maxpool3x3s2_idx: C must be a multiple of the 16-byte vector");
  const int64_t n = (int64_t)B * H * W * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(maxpool3s2_bwd_idx_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const unsigned char*)idx, (const bf16_t*)dy, (bf16_t*)dx, accumulate, B, H, W, C),
          hipLaunchKernelGGL(maxpool3s2_bwd_idx_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const unsigned char*)idx, (const float*)dy, (float*)dx, accumulate, B, H, W, C));
  return ksmi_check_launch("maxpool3x3s2_bwd_idx");
}

int ksmi_affine(const void* x, const float* scale, const float* shift, void* y, int64_t npix, int C, int relu, float alpha, int dtype,
                void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || (scale == nullptr) != (shift == nullptr)) return ksmi_fail(KSMI_E_ARG, "affine: bad args");
  const int64_t nvec = npix * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(affine_kernel<bf16_t>, dim3(grid_for(nvec, 65536)), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, nvec, C / vec, relu, alpha),
          hipLaunchKernelGGL(affine_kernel<float>, dim3(grid_for(nvec, 65536)), dim3(256), 0, st, (const float*)x, scale, shift, (float*)y, nvec, C / vec, relu, alpha));
  return ksmi_check_launch("affine");
}

int ksmi_col2im(const void* dcol, void* dx, int accumulate, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                int Kpad, int dtype, void* stream) {
  const int64_t n = (int64_t)B * H * W * Cin;
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)dcol, (bf16_t*)dx, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad, accumulate),
          hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)dcol, (float*)dx, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, Kpad, accumulate));
  return ksmi_check_launch("col2im");
}

struct DwRowGeom { int seg, nseg; int64_t units; };
static DwRowGeom dw_row_geom(int B, int H, int W) {
  static const int want = ksmi_knob_int("KSMI_DW_SEG", 14);
  DwRowGeom q;
  q.nseg = W <= want ? 1 : (W + want / 2) / want;
  q.seg = (W + q.nseg - 1) / q.nseg;
  q.nseg = (W + q.seg - 1) / q.seg;
  q.units = (int64_t)B * H * q.nseg;
  return q;
}
static bool dw_row_form() {
  static const bool on = (ksmi_knob_int("KSMI_DW_ROW", 1) != 0);
  return on;
}

int ksmi_dwconv3x3_gelu_forward(const void* x, const float* w, const float* bias, void* z, void* g, int B, int H, int W, int C,
                                int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "dwconv: C must be a multiple of the 16-byte vector");
  hipStream_t st = (hipStream_t)stream;
  if (dw_row_form()) {
    const DwRowGeom q = dw_row_geom(B, H, W);
    const dim3 grid((unsigned)(((q.units + 7) / 8 + 7) / 8 * 8), (C / vec + 31) / 32);
    KSMI_DT(dtype,
            hipLaunchKernelGGL((dwconv3x3_row_kernel<bf16_t, 0>), grid, dim3(256), 0, st, (const bf16_t*)x, w, bias, (bf16_t*)z, (bf16_t*)g, B, H, W, C, q.seg, q.nseg, q.units),
            hipLaunchKernelGGL((dwconv3x3_row_kernel<float, 0>), grid, dim3(256), 0, st, (const float*)x, w, bias, (float*)z, (float*)g, B, H, W, C, q.seg, q.nseg, q.units));
    return ksmi_check_launch("dwconv3x3_gelu_fwd");
  }
  static const int ppb_env = ksmi_knob_int("KSMI_DW_PPB", 0);
  const int ppb = ppb_env > 0 ? ppb_env : 32;       // measured (KSMI_DW_PPB sweep): 128 -> 110 / 69 us, 32 -> 90 / 49 us (forward / adjoint)
  const dim3 grid((unsigned)((((int64_t)B * H * W + ppb - 1) / ppb + 7) / 8 * 8), (C / vec + 31) / 32);
  KSMI_DT(dtype,
          hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t, 0>), grid, dim3(256), 0, st, (const bf16_t*)x, w, bias, (bf16_t*)z, (bf16_t*)g, B, H, W, C, ppb),
          hipLaunchKernelGGL((dwconv3x3_kernel<float, 0>), grid, dim3(256), 0, st, (const float*)x, w, bias, (float*)z, (float*)g, B, H, W, C, ppb));
  return ksmi_check_launch("dwconv3x3_gelu_fwd");
}

int ksmi_dropout_apply(const void* x, const void* resid, void* y, int64_t rows, int cols, int rows_per_sample, uint32_t thr, float inv_keep,
                       uint32_t site, uint32_t dp_thr, float dp_inv_keep, uint32_t dp_site, const uint32_t* rng_state, int dtype, void* stream);

int ksmi_dwconv3x3_gelu_forward_drop(const void* x, const float* w, const float* bias, void* z, void* g, int B, int H, int W, int C, uint32_t thr,
                                     float inv_keep, uint32_t site, const uint32_t* rng_state, int dtype, void* stream) {
  if (!thr) return ksmi_dwconv3x3_gelu_forward(x, w, bias, z, g, B, H, W, C, dtype, stream);
  if (!rng_state) return ksmi_fail(KSMI_E_ARG, "dwconv3x3_gelu_forward_drop: the rng state is required");
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "dwconv: C must be a multiple of the 16-byte vector");
  if ((int64_t)B * H * W * C >= ((int64_t)1 << 32)) return ksmi_fail(KSMI_E_ARG, "dwconv3x3_gelu_forward_drop: element index exceeds 32 bits");
  if (!dw_row_form()) {            // (the per-pixel form of KSMI_DW_ROW=0: the two passes it replaces)
    if (int rc = ksmi_dwconv3x3_gelu_forward(x, w, bias, z, g, B, H, W, C, dtype, stream)) return rc;
    return ksmi_dropout_apply(g, nullptr, g, (int64_t)B * H * W, C, H * W, thr, inv_keep, site, 0, 1.f, 0, rng_state, dtype, stream);
  }
  hipStream_t st = (hipStream_t)stream;
  const DwRowGeom q = dw_row_geom(B, H, W);
  const dim3 grid((unsigned)(((q.units + 7) / 8 + 7) / 8 * 8), (C / vec + 31) / 32);
  KSMI_DT(dtype,
          hipLaunchKernelGGL((dwconv3x3_row_kernel<bf16_t, 0, true>), grid, dim3(256), 0, st, (const bf16_t*)x, w, bias, (bf16_t*)z, (bf16_t*)g, B, H, W, C, q.seg, q.nseg, q.units, thr, inv_keep, site, rng_state),
          hipLaunchKernelGGL((dwconv3x3_row_kernel<float, 0, true>), grid, dim3(256), 0, st, (const float*)x, w, bias, (float*)z, (float*)g, B, H, W, C, q.seg, q.nseg, q.units, thr, inv_keep, site, rng_state));
  return ksmi_check_launch("dwconv3x3_gelu_fwd_drop");
}

int ksmi_dwconv3x3_backward_input(const void* dz, const float* w, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "dwconv: C must be a multiple of the 16-byte vector");
  hipStream_t st = (hipStream_t)stream;
  if (dw_row_form()) {
    const DwRowGeom q = dw_row_geom(B, H, W);
    const dim3 grid((unsigned)(((q.units + 7) / 8 + 7) / 8 * 8), (C / vec + 31) / 32);
    KSMI_DT(dtype,
            hipLaunchKernelGGL((dwconv3x3_row_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)dz, w, (const float*)nullptr, (bf16_t*)dx, (bf16_t*)nullptr, B, H, W, C, q.seg, q.nseg, q.units),
            hipLaunchKernelGGL((dwconv3x3_row_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)dz, w, (const float*)nullptr, (float*)dx, (float*)nullptr, B, H, W, C, q.seg, q.nseg, q.units));
    return ksmi_check_launch("dwconv3x3_bwd_input");
  }
  static const int ppb_env = ksmi_knob_int("KSMI_DW_PPB", 0);
  const int ppb = ppb_env > 0 ? ppb_env : 32;       // measured (KSMI_DW_PPB sweep): 128 -> 110 / 69 us, 32 -> 90 / 49 us (forward / adjoint)
  const dim3 grid((unsigned)((((int64_t)B * H * W + ppb - 1) / ppb + 7) / 8 * 8), (C / vec + 31) / 32);
  KSMI_DT(dtype,
          hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t, 1>), grid, dim3(256), 0, st, (const bf16_t*)dz, w, (const float*)nullptr, (bf16_t*)dx, (bf16_t*)nullptr, B, H, W, C, ppb),
          hipLaunchKernelGGL((dwconv3x3_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)dz, w, (const float*)nullptr, (float*)dx, (float*)nullptr, B, H, W, C, ppb));
  return ksmi_check_launch("dwconv3x3_bwd_input");
}

int ksmi_dwconv3x3_wgrad(const void* x, const void* dz, float* partial, int rows, int B, int H, int W, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || rows < 1) return ksmi_fail(KSMI_E_ARG, "dwconv_wgrad: bad args");
  const int cvb = C / vec <= 32 ? 32 : 64;
  const dim3 grid(rows, (C / vec + cvb - 1) / cvb);
  hipStream_t st = (hipStream_t)stream;
  if (dw_row_form()) {
    const DwRowGeom q = dw_row_geom(B, H, W);
    if (cvb == 32)
      KSMI_DT(dtype,
              hipLaunchKernelGGL((dwconv3x3_wgrad_row_kernel<bf16_t, 32>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dz, partial, B, H, W, C, q.seg, q.nseg, q.units),
              hipLaunchKernelGGL((dwconv3x3_wgrad_row_kernel<float, 32>), grid, dim3(256), 0, st, (const float*)x, (const float*)dz, partial, B, H, W, C, q.seg, q.nseg, q.units));
    else
      KSMI_DT(dtype,
              hipLaunchKernelGGL((dwconv3x3_wgrad_row_kernel<bf16_t, 64>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dz, partial, B, H, W, C, q.seg, q.nseg, q.units),
              hipLaunchKernelGGL((dwconv3x3_wgrad_row_kernel<float, 64>), grid, dim3(256), 0, st, (const float*)x, (const float*)dz, partial, B, H, W, C, q.seg, q.nseg, q.units));
    return ksmi_check_launch("dwconv3x3_wgrad");
  }
  if (cvb == 32)
    KSMI_DT(dtype,
            hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<bf16_t, 32>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dz, partial, B, H, W, C),
            hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<float, 32>), grid, dim3(256), 0, st, (const float*)x, (const float*)dz, partial, B, H, W, C));
  else
    KSMI_DT(dtype,
            hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<bf16_t, 64>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dz, partial, B, H, W, C),
            hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<float, 64>), grid, dim3(256), 0, st, (const float*)x, (const float*)dz, partial, B, H, W, C));
  return ksmi_check_launch("dwconv3x3_wgrad");
}

static bool sr_mfma(int dtype, int Nk, int C, int H) {
  static const bool valu = ksmi_knob_is_set("KSMI_ATTN_VALU");
  return dtype == KSMI_BF16 && Nk <= 64 && H > 0 && (C / H == 64 || C / H == 80) && !valu;
}

int ksmi_sr_attention_forward_drop(const void* q, const void* kv, void* out, int B, int Nq, int Nk, int H, int C, float scale,
                                   uint32_t drop_thr, float drop_inv_keep, uint32_t site, const uint32_t* rng_state, int dtype, void* stream) {
  if (drop_thr && !rng_state) return ksmi_fail(KSMI_E_ARG, "sr_attention: dropout needs the rng state");
  if ((uint64_t)B * H * Nq * Nk >= (1ull << 32)) return ksmi_fail(KSMI_E_UNSUPPORTED, "sr_attention: B*H*Nq*Nk must stay below 2^32");
  if (sr_mfma(dtype, Nk, C, H))
    return ksmi_attn_mfma_sr(0, q, kv, out, nullptr, nullptr, nullptr, nullptr, nullptr, B, Nq, Nk, H, C, scale, drop_thr, drop_inv_keep, site,
                             rng_state, stream);
  const SrDrop dr = {drop_thr, site, drop_inv_keep, rng_state};
  return sr_attn_dispatch(0, q, kv, nullptr, out, nullptr, nullptr, nullptr, B, Nq, Nk, H, C, scale, 1, dtype, dr, (hipStream_t)stream);
}
int ksmi_sr_attention_forward(const void* q, const void* kv, void* out, int B, int Nq, int Nk, int H, int C, float scale, int dtype,
                              void* stream) {
  return ksmi_sr_attention_forward_drop(q, kv, out, B, Nq, Nk, H, C, scale, 0u, 1.f, 0u, nullptr, dtype, stream);
}

size_t ksmi_sr_attention_bwd_workspace_valu(int B, int Nq, int Nk, int H, int C);
int ksmi_sr_attention_splits(int Nq) { int s = (Nq + 255) / 256; return s < 1 ? 1 : (s > 16 ? 16 : s); }

size_t ksmi_sr_attention_bwd_workspace(int B, int Nq, int Nk, int H, int C) {
  const size_t mf = H > 0 ? ksmi_attn_mfma_workspace(B, Nq, Nk, H, C / H) : 0;
  const size_t vl = ksmi_sr_attention_bwd_workspace_valu(B, Nq, Nk, H, C);
  return mf > vl ? mf : vl;
}

size_t ksmi_sr_attention_bwd_workspace_valu(int B, int Nq, int Nk, int H, int C) {
  const size_t ps = (size_t)B * H * Nq * Nk * sizeof(float);
  return 2 * ps + (size_t)ksmi_sr_attention_splits(Nq) * B * Nk * 2 * C * sizeof(float);
}

int ksmi_sr_attention_backward_drop(const void* q, const void* kv, const void* out, const void* dout, void* dq, void* dkv, void* workspace,
                                    int B, int Nq, int Nk, int H, int C, float scale, uint32_t drop_thr, float drop_inv_keep, uint32_t site,
                                    const uint32_t* rng_state, int dtype, void* stream) {
  if (drop_thr && !rng_state) return ksmi_fail(KSMI_E_ARG, "sr_attention: dropout needs the rng state");
  if (sr_mfma(dtype, Nk, C, H))
    return ksmi_attn_mfma_sr(1, q, kv, (void*)out, nullptr, dout, dq, dkv, workspace, B, Nq, Nk, H, C, scale, drop_thr, drop_inv_keep, site,
                             rng_state, stream);
  const SrDrop dr = {drop_thr, site, drop_inv_keep, rng_state};
  const size_t ps = (size_t)B * H * Nq * Nk;
  float* Pb = (float*)workspace;
  float* Sb = Pb + ps;
  float* partial = Sb + ps;
  const int nsplit = ksmi_sr_attention_splits(Nq);
  hipStream_t st = (hipStream_t)stream;
  int rc = sr_attn_dispatch(1, q, kv, dout, dq, Pb, Sb, partial, B, Nq, Nk, H, C, scale, nsplit, dtype, dr, st);
  if (rc) return rc;
  rc = sr_attn_dispatch(2, q, kv, dout, nullptr, Pb, Sb, partial, B, Nq, Nk, H, C, scale, nsplit, dtype, dr, st);
  if (rc) return rc;
  const int64_t n = (int64_t)B * Nk * 2 * C;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(sum_splits_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, partial, (bf16_t*)dkv, n, nsplit),
          hipLaunchKernelGGL(sum_splits_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, partial, (float*)dkv, n, nsplit));
  return ksmi_check_launch("sr_attention_sum");
}
int ksmi_sr_attention_backward(const void* q, const void* kv, const void* out, const void* dout, void* dq, void* dkv, void* workspace, int B,
                               int Nq, int Nk, int H, int C, float scale, int dtype, void* stream) {
  return ksmi_sr_attention_backward_drop(q, kv, out, dout, dq, dkv, workspace, B, Nq, Nk, H, C, scale, 0u, 1.f, 0u, nullptr, dtype, stream);
}

int ksmi_bilinear_forward(const void* x, const void* add, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "bilinear: C must be a multiple of the 16-byte vector");
  const int64_t n = (int64_t)B * Ho * Wo * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)add, (bf16_t*)y, B, Hi, Wi, Ho, Wo, C),
          hipLaunchKernelGGL(bilinear_fwd_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)x, (const float*)add, (float*)y, B, Hi, Wi, Ho, Wo, C));
  return ksmi_check_launch("bilinear_fwd");
}

int ksmi_bilinear_backward(const void* dy, void* dx, int accumulate, int B, int Hi, int Wi, int Ho, int Wo, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || Ho < Hi || Wo < Wi) return ksmi_fail(KSMI_E_ARG, "bilinear_bwd: upsampling only, C multiple of the vector");
  const int64_t n = (int64_t)B * Hi * Wi * (C / vec);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bilinear_bwd_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, Hi, Wi, Ho, Wo, C, accumulate),
          hipLaunchKernelGGL(bilinear_bwd_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)dy, (float*)dx, B, Hi, Wi, Ho, Wo, C, accumulate));
  return ksmi_check_launch("bilinear_bwd");
}

int ksmi_bn_bwd_apply(const void* dy, const void* r, const float* mean, const float* rstd, const float* gamma, const float* sums,
                      void* dv, int relu_mask, double count, int64_t npix, int C, int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec) return ksmi_fail(KSMI_E_ARG, "bn_bwd_apply: C must be a multiple of the 16-byte vector");
  const int64_t nvec = npix * (C / vec);
  const float inv_n = (float)(1.0 / count);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(grid_for(nvec, 65536)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)r, mean, rstd, gamma, sums, (bf16_t*)dv, relu_mask, inv_n, nvec, C / vec, C),
          hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid_for(nvec, 65536)), dim3(256), 0, st, (const float*)dy, (const float*)r, mean, rstd, gamma, sums, (float*)dv, relu_mask, inv_n, nvec, C / vec, C));
  return ksmi_check_launch("bn_bwd_apply");
}

int ksmi_bn_bwd_reduce(const void* dy, const void* x, const float* mean, const float* rstd, float* partial, int rows, int64_t npix, int C,
                       int dtype, void* stream) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  if (C % vec || rows < 1) return ksmi_fail(KSMI_E_ARG, "bn_bwd_reduce: bad args");
  const dim3 grid(rows, (C / vec + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, partial, npix, C),
          hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, (const float*)x, mean, rstd, partial, npix, C));
  return ksmi_check_launch("bn_bwd_reduce");
}

int ksmi_sar_preprocess(const float* x, const float* mean, const float* stdv, float* y, int B, int C, int64_t HW, float clamp_input, void* stream) {
  const int64_t n = (int64_t)B * C * HW;
  hipLaunchKernelGGL(sar_preprocess_kernel, dim3(grid_for(n, 65536)), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, y, C, HW, n, clamp_input);
  return ksmi_check_launch("sar_preprocess");
}

int ksmi_out_to_nchw(const void* x, float* y, int B, int C, int Cs, int64_t HW, int act, int dtype, void* stream) {
  const int64_t n = (int64_t)B * C * HW;
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(out_to_nchw_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const bf16_t*)x, y, B, C, Cs, HW, act),
          hipLaunchKernelGGL(out_to_nchw_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, (const float*)x, y, B, C, Cs, HW, act));
  return ksmi_check_launch("out_to_nchw");
}

int ksmi_dout_to_nhwc(const float* dy, const float* y, void* dx, int B, int C, int Cs, int64_t HW, int act, int dtype, void* stream) {
  const int64_t n = (int64_t)B * HW * Cs;
  hipStream_t st = (hipStream_t)stream;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(dout_to_nhwc_kernel<bf16_t>, dim3(grid_for(n, 65536)), dim3(256), 0, st, dy, y, (bf16_t*)dx, B, C, Cs, HW, act),
          hipLaunchKernelGGL(dout_to_nhwc_kernel<float>, dim3(grid_for(n, 65536)), dim3(256), 0, st, dy, y, (float*)dx, B, C, Cs, HW, act));
  return ksmi_check_launch("dout_to_nhwc");
}

}  // extern "C"
