// Shared epilogue of the implicit-GEMM kernels (see igemm.hip / igemm2.hip).
#pragma once
#include "common.h"
#include "../../include/ksmi.h"

// LDS swizzle of the 16-byte k-group slot inside a 64-byte row (conflict-free
// ds_read_b128 for 16 consecutive rows; derivation in DESIGN.md §LDS).
// (By the lane groups MI355X_MICROARCH.md lists for ds_read_b128 -- {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- this choice is 2-way
// conflicted for odd halo alignments and h(q) = ((q & 1) << 1) would be conflict-free for all; measured on the long-K and short-K
// convolutions the two are equal within noise (profiles/midconv_probe.py), so the K loop is not LDS-bank bound and this one stays.)
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }

// Exact unsigned division by a runtime constant with one v_mul_hi_u32: q = umulhi(p, floor(2^32/d)+1)
// (exact for p*d < 2^32; all uses have p <= 2048, d <= 512).  d == 1 is handled by the caller's struct.
__host__ __device__ inline uint32_t fastdiv_magic(int dd) { return dd > 1 ? (uint32_t)(0x100000000ull / (uint64_t)dd) + 1u : 0u; }
struct FastDiv {
  uint32_t magic; int d;
  __device__ __forceinline__ explicit FastDiv(int dd) : magic(dd > 1 ? (uint32_t)(4294967296.0 / (double)dd) + 1u : 0u), d(dd) {}
  __device__ __forceinline__ FastDiv(int dd, uint32_t host_magic) : magic(host_magic), d(dd) {}   // magic from the launcher
  __device__ __forceinline__ int div(int p) const { return d > 1 ? (int)__umulhi((uint32_t)p, magic) : p; }
};

// k-chunk table lookups as 32-bit scalar loads: a dynamically indexed uint8/uint16 kernarg array is
// fetched with global_load_ubyte/ushort (VMEM), and hipcc then waits vmcnt(0) -- draining every LDS-DMA
// in flight -- before the value can be used.  A uniform dword index keeps it on the scalar path.
__device__ __forceinline__ int chunk_c0_of(const ksmi_conv_desc& d, int ch) {
  if (d.uniform_kc) return ch * d.uniform_kc;
  const uint32_t w = ((const uint32_t*)d.chunk_c0)[ch >> 1];
  return (ch & 1) ? (int)(w >> 16) : (int)(w & 0xffffu);
}
__device__ __forceinline__ int chunk_src_of(const ksmi_conv_desc& d, int ch) {
  if (d.uniform_kc) return 0;
  return (int)((((const uint32_t*)d.chunk_src)[ch >> 2] >> ((ch & 3) * 8)) & 0xffu);
}

// source pixel index of logical input pixel (b, iy, ix): dense, or the strided view of a parity sub-image
__device__ __forceinline__ int src_pixel(const ksmi_conv_desc& d, int b, int iy, int ix) {
  if (d.in_sy == 0) return (b * d.Hin + iy) * d.Win + ix;
  return (b * d.in_H + (iy * d.in_sy + d.in_oy)) * d.in_W + (ix * d.in_sx + d.in_ox);
}

// destination pixel index of result pixel (b, oy, ox): dense, or the strided placement of a phase convolution
__device__ __forceinline__ size_t dst_pixel(const ksmi_conv_desc& d, int b, int oy, int ox) {
  if (d.out_sy == 0) return ((size_t)b * d.Hout + oy) * d.Wout + ox;
  return ((size_t)b * d.out_H + (oy * d.out_sy + d.out_oy)) * d.out_W + (ox * d.out_sx + d.out_ox);
}

template <typename T, int NT>
__device__ __forceinline__ void igemm_epilogue(const ksmi_conv_desc& d, f32x4 (&acc)[4][NT], unsigned char* smem, int tid,
                                               int wave, int g, int l15, int b, int oy0, int ox0, int n0, int P) {
  constexpr int BN = NT * 16;
  constexpr int VEC = ElemTraits<T>::kVec;
  const FastDiv dTW(d.TW);
  // ---- epilogue ---------------------------------------------------------------------------------
  // phase 1: accumulators (+bias) -> LDS tile [P][BN] in T (C layout: col n = l15, row = g*4 + r);
  //          plain BatchNorm statistics (sum, sumsq) of the values as stored (rounded to T), from registers.
  // phase 2: 16-byte vectors LDS -> (mask, accumulate) -> coalesced global stores.
  constexpr int LDT = BN + 16 / (int)sizeof(T);          // row stride in elements (+16 B pad)
  T* tile = (T*)smem;
  float s_sum[NT], s_sq[NT];
#pragma unroll
  for (int nf = 0; nf < NT; ++nf) { s_sum[nf] = 0.f; s_sq[nf] = 0.f; }
  bool rvalid[4][4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = wave * 64 + mf * 16 + g * 4 + r;
      const int ly = dTW.div(p), lx = p - ly * d.TW;
      rvalid[mf][r] = p < P && (oy0 + ly) < d.Hout && (ox0 + lx) < d.Wout;
    }
  __syncthreads();                                       // every wave is done reading halo / weights
#pragma unroll
  for (int nf = 0; nf < NT; ++nf) {
    const int n = n0 + nf * 16 + l15;
    const bool nvalid = n < d.N;
    const float bias = (d.bias && nvalid) ? d.bias[d.ps_cout > 0 ? n % d.ps_cout : n] : 0.f;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = wave * 64 + mf * 16 + g * 4 + r;
        const float v = acc[mf][nf][r] + bias;
        const float q = ElemTraits<T>::cvt(v);               // statistics of the STORED value (identity in fp32), round 5
        if (rvalid[mf][r] && nvalid) { s_sum[nf] += q; s_sq[nf] += q * q; }
        if (p < P) ElemTraits<T>::st(tile + p * LDT + nf * 16 + l15, v);
      }
  }
  __syncthreads();
  constexpr int VPR = BN / VEC;                          // vectors per tile row
  const int myv = tid % VPR;                             // 256 % VPR == 0: fixed channel-vector per thread
  const int nq = n0 + myv * VEC;
  float m_sum[VEC], m_sq[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { m_sum[j] = 0.f; m_sq[j] = 0.f; }
  if (nq < d.N) {
    int si = 0;
    for (int k = 1; k < d.ndst; ++k) if (nq >= d.dst[k].n_begin) si = k;
    const ksmi_dst& ds = d.dst[si];
    const int nval = min(VEC, d.N - nq);
    const bool vec_ok = nval == VEC && ((ds.C | ds.c_off | (nq - ds.n_begin)) % VEC) == 0 && (d.N % VEC) == 0 &&
                        (d.ps_cout == 0 || d.ps_cout % VEC == 0);
    float mm[VEC], mr[VEC], mg[VEC], mb[VEC];
    if (d.mask_src) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int n = min(nq + j, d.N - 1);
        mm[j] = d.m_mean[n]; mr[j] = d.m_rstd[n]; mg[j] = d.m_scale[n]; mb[j] = d.m_shift[n];
      }
    }
    for (int p = tid / VPR; p < P; p += 256 / VPR) {
      const int ly = dTW.div(p), lx = p - ly * d.TW;
      const int oy = oy0 + ly, ox = ox0 + lx;
      if (oy >= d.Hout || ox >= d.Wout) continue;
      float v[VEC];
      vec_unpack<T>(*(const u32x4*)(tile + p * LDT + myv * VEC), v);
      const size_t opix = ((size_t)b * d.Hout + oy) * d.Wout + ox;
      if (d.mask_src) {
        float m[VEC];
        const T* mp = (const T*)d.mask_src + opix * d.N + nq;
        if (vec_ok) vec_unpack<T>(*(const u32x4*)mp, m);
        else for (int j = 0; j < VEC; ++j) m[j] = j < nval ? ElemTraits<T>::ld(mp + j) : 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xh = (m[j] - mm[j]) * mr[j];
          if (!(m[j] * mg[j] + mb[j] > 0.f)) v[j] = 0.f;
          if (j < nval) { m_sum[j] += v[j]; m_sq[j] += v[j] * xh; }
        }
      }
      T* dp;
      if (d.ps_cout > 0) {
        const int dd = nq / d.ps_cout, nn = nq - dd * d.ps_cout;
        const size_t op2 = ((size_t)b * (2 * d.Hout) + (2 * oy + (dd >> 1))) * (2 * d.Wout) + (2 * ox + (dd & 1));
        dp = (T*)ds.ptr + op2 * ds.C + ds.c_off + nn;
      } else {
        dp = (T*)ds.ptr + dst_pixel(d, b, oy, ox) * ds.C + ds.c_off + (nq - ds.n_begin);
      }
      if (vec_ok) {
        if (ds.accumulate) {
          float o[VEC];
          vec_unpack<T>(*(const u32x4*)dp, o);
#pragma unroll
          for (int j = 0; j < VEC; ++j) v[j] += o[j];
        }
        *(u32x4*)dp = vec_pack<T>(v);
      } else {
        for (int j = 0; j < nval; ++j) {
          // scalar tail: per-column segment lookup (ps_cout columns stay inside one quadrant only when aligned)
          const int n = nq + j;
          int sj = 0;
          for (int k = 1; k < d.ndst; ++k) if (n >= d.dst[k].n_begin) sj = k;
          const ksmi_dst& dj = d.dst[sj];
          T* q;
          if (d.ps_cout > 0) {
            const int dd = n / d.ps_cout, nn = n - dd * d.ps_cout;
            const size_t op2 = ((size_t)b * (2 * d.Hout) + (2 * oy + (dd >> 1))) * (2 * d.Wout) + (2 * ox + (dd & 1));
            q = (T*)dj.ptr + op2 * dj.C + dj.c_off + nn;
          } else {
            q = (T*)dj.ptr + dst_pixel(d, b, oy, ox) * dj.C + dj.c_off + (n - dj.n_begin);
          }
          float o = v[j];
          if (dj.accumulate) o += ElemTraits<T>::ld(q);
          ElemTraits<T>::st(q, o);
        }
      }
    }
  }
  if (d.stats) {
    __syncthreads();                                     // tile reads finished; reuse LDS for the reduction
    float* red = (float*)smem;
    if (d.mask_src) {
      // [256 threads][2] per channel j of the vector, reduced over the threads that share myv
      for (int j = 0; j < VEC; ++j) {
        __syncthreads();
        red[tid * 2 + 0] = m_sum[j]; red[tid * 2 + 1] = m_sq[j];
        __syncthreads();
        if (tid < VPR * 2) {
          const int q = tid >> 1, which = tid & 1;
          float a = 0.f;
          for (int t = q; t < 256; t += VPR) a += red[t * 2 + which];
          const int n = n0 + q * VEC + j;
          if (n < d.Npad) d.stats[((size_t)blockIdx.x * 2 + which) * d.Npad + n] = a;
        }
      }
    } else {
#pragma unroll
      for (int nf = 0; nf < NT; ++nf) {
        float a = s_sum[nf], q = s_sq[nf];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (g == 0) { red[(wave * 2 + 0) * BN + nf * 16 + l15] = a; red[(wave * 2 + 1) * BN + nf * 16 + l15] = q; }
      }
      __syncthreads();
      if (tid < 2 * BN) {
        const int which = tid / BN, n = tid - which * BN;
        const float v = red[(0 * 2 + which) * BN + n] + red[(1 * 2 + which) * BN + n] + red[(2 * 2 + which) * BN + n] + red[(3 * 2 + which) * BN + n];
        if (n0 + n < d.Npad) d.stats[((size_t)blockIdx.x * 2 + which) * d.Npad + n0 + n] = v;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Direct epilogue for the swapped-operand MFMA (D = W * X^T): acc[mf][nf][r] holds
//   out[pixel = wave*64 + mf*16 + l15][n = n0 + nf*16 + g*4 + r]
// i.e. every lane owns 4 CONSECUTIVE channels of a pixel -> one 8-byte (bf16) / 16-byte (fp32) global
// access per (mf, nf), no LDS staging and no barrier.  Same features as igemm_epilogue.
// -------------------------------------------------------------------------------------------------
template <typename T> struct Quad;   // 4 consecutive elements of T as one vector access
template <> struct Quad<float> {
  __device__ static __forceinline__ void ld(const float* p, float* f) { const f32x4 v = *(const f32x4*)p; f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
  __device__ static __forceinline__ void st(float* p, const float* f) { *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]}; }
};
template <> struct Quad<bf16_t> {
  __device__ static __forceinline__ void ld(const bf16_t* p, float* f) {
    const uint2 v = *(const uint2*)p;
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void st(bf16_t* p, const float* f) {
    uint2 v;
    v.x = (uint32_t)f32_to_bf16(f[0]) | ((uint32_t)f32_to_bf16(f[1]) << 16);
    v.y = (uint32_t)f32_to_bf16(f[2]) | ((uint32_t)f32_to_bf16(f[3]) << 16);
    *(uint2*)p = v;
  }
};

// sum over the 16 lanes of a DPP row (lane & 15), result in every lane: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
  return v;
}

// Channel ownership of the direct epilogue.  The weight slab is staged with its rows permuted (igemm2.hip: row nf*16 + g*4 + r of
// the MFMA A operand holds output channel g*4*NT + nf*4 + r of the tile), so lane (g, l15) owns the 4*NT CONSECUTIVE channels
// n0 + g*4*NT .. of pixel l15: one 16-byte store per pixel for bf16 NT = 2 (a wave instruction covers 16 whole 64-byte pixel rows
// instead of 16 half rows).
template <int NT> __device__ __forceinline__ int epi_col(int nf, int g) { return g * 4 * NT + nf * 4; }

// wave = pixel group (0..3) of the calling wave; n0 = first channel of the WAVE; wn / n0wg = channel group of the wave and
// first channel of the workgroup (WN channel groups share the statistics reduction buffer)
template <typename T, int NT, bool EX = true, int WN = 1>
__device__ __forceinline__ void igemm_epilogue_direct(const ksmi_conv_desc& d, f32x4 (&acc)[4][NT], unsigned char* smem, int tid,
                                                      int wave, int g, int l15, int b, int oy0, int ox0, int n0, int P,
                                                      uint32_t magic_tw = 0xffffffffu, int wn = 0, int n0wg = -1, int stats_row = -1) {
  constexpr int BN = NT * 16;
  const FastDiv dTW = magic_tw != 0xffffffffu ? FastDiv(d.TW, magic_tw) : FastDiv(d.TW);
  size_t opix[4];
  bool pv[4];
  int oyv[4], oxv[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    const int p = wave * 64 + mf * 16 + l15;
    const int ly = dTW.div(p), lx = p - ly * d.TW;
    oyv[mf] = oy0 + ly; oxv[mf] = ox0 + lx;
    pv[mf] = p < P && oyv[mf] < d.Hout && oxv[mf] < d.Wout;
    opix[mf] = ((size_t)b * d.Hout + oyv[mf]) * d.Wout + oxv[mf];
  }
  // ---- per-nf column setup ---------------------------------------------------------------------------
  float ssum[NT][4], ssq[NT][4], bias[NT][4];
  int n4v[NT], nval[NT], si[NT], ddv[NT], nnv[NT];
  bool vok[NT];
#pragma unroll
  for (int nf = 0; nf < NT; ++nf) {
    const int n4 = n0 + epi_col<NT>(nf, g);
    n4v[nf] = n4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[nf][r] = 0.f; ssq[nf][r] = 0.f; bias[nf][r] = 0.f; }
    nval[nf] = n4 >= d.N ? 0 : min(4, d.N - n4);
    int k0 = 0;
    for (int k = 1; k < d.ndst; ++k) if (n4 >= d.dst[k].n_begin) k0 = k;
    si[nf] = k0;
    const ksmi_dst& ds = d.dst[k0];
    vok[nf] = nval[nf] == 4 && ((ds.C | ds.c_off | (n4 - ds.n_begin) | d.N) & 3) == 0 && (d.ps_cout & 3) == 0;
    if (d.bias)
      for (int r = 0; r < nval[nf]; ++r) bias[nf][r] = d.bias[d.ps_cout > 0 ? (n4 + r) % d.ps_cout : (n4 + r)];
    ddv[nf] = 0; nnv[nf] = n4 - ds.n_begin;
    if (d.ps_cout > 0) { ddv[nf] = n4 / d.ps_cout; nnv[nf] = n4 - ddv[nf] * d.ps_cout; }
  }
  // 16-byte combined store of the lane's 8 consecutive bf16 channels (NT = 2): same destination, 8-element alignment
  bool wide = false;
  if constexpr (NT == 2 && sizeof(T) == 2) {
    const ksmi_dst& ds = d.dst[si[0]];
    wide = vok[0] && vok[1] && si[0] == si[1] && ddv[0] == ddv[1] && (((ds.C | ds.c_off | nnv[0]) & 7) == 0) && !ds.accumulate;
  }
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    if (!pv[mf]) continue;
    float v[NT][4];
#pragma unroll
    for (int nf = 0; nf < NT; ++nf) {
      const int n4 = n4v[nf];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nf][r] = acc[mf][nf][r] + bias[nf][r];
      if (nval[nf] == 0) continue;
      if (EX && d.alpha != 0.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nf][r] *= d.alpha;
      }
      if (EX && d.resid) {
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
        const T* rp = (const T*)d.resid + opix[mf] * d.residC + n4;
        if (vok[nf] && (d.residC & 3) == 0) Quad<T>::ld(rp, rs);
        else for (int r = 0; r < nval[nf]; ++r) rs[r] = ElemTraits<T>::ld(rp + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nf][r] += rs[r];
      }
      if (EX && d.relu_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nf][r] = fmaxf(v[nf][r], 0.f);
      }
      if (d.mask_src) {
        float m[4] = {0.f, 0.f, 0.f, 0.f};
        const T* mp = (const T*)d.mask_src + opix[mf] * d.N + n4;
        if (vok[nf]) Quad<T>::ld(mp, m);
        else for (int r = 0; r < nval[nf]; ++r) m[r] = ElemTraits<T>::ld(mp + r);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = min(n4 + r, d.N - 1);
          const float xh = (m[r] - d.m_mean[n]) * d.m_rstd[n];
          if (!(m[r] * d.m_scale[n] + d.m_shift[n] > 0.f)) v[nf][r] = 0.f;
          const float q = ElemTraits<T>::cvt(v[nf][r]);              // statistics of the STORED value (identity in fp32), round 5
          if (r < nval[nf]) { ssum[nf][r] += q; ssq[nf][r] += q * xh; }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (r < nval[nf]) { const float q = ElemTraits<T>::cvt(v[nf][r]); ssum[nf][r] += q; ssq[nf][r] += q * q; }
      }
    }
    // ---- stores -----------------------------------------------------------------------------------
    auto dst_of = [&](const ksmi_dst& ds, int dd, int nn) -> T* {
      if (d.ps_cout > 0) {
        const size_t op2 = ((size_t)b * (2 * d.Hout) + (2 * oyv[mf] + (dd >> 1))) * (2 * d.Wout) + (2 * oxv[mf] + (dd & 1));
        return (T*)ds.ptr + op2 * ds.C + ds.c_off + nn;
      }
      return (T*)ds.ptr + (EX ? dst_pixel(d, b, oyv[mf], oxv[mf]) : opix[mf]) * ds.C + ds.c_off + nn;
    };
    if constexpr (NT == 2 && sizeof(T) == 2) {
      if (wide) {
        float f[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { f[r] = v[0][r]; f[4 + r] = v[1][r]; }
        *(u32x4*)dst_of(d.dst[si[0]], ddv[0], nnv[0]) = vec_pack<T>(f);
        continue;
      }
    }
#pragma unroll
    for (int nf = 0; nf < NT; ++nf) {
      if (nval[nf] == 0) continue;
      const ksmi_dst& ds = d.dst[si[nf]];
      if (vok[nf]) {
        T* dp = dst_of(ds, ddv[nf], nnv[nf]);
        if (ds.accumulate) {
          float o[4];
          Quad<T>::ld(dp, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[nf][r] += o[r];
        }
        Quad<T>::st(dp, v[nf]);
      } else {
        for (int r = 0; r < nval[nf]; ++r) {
          const int n = n4v[nf] + r;
          int sj = 0;
          for (int k = 1; k < d.ndst; ++k) if (n >= d.dst[k].n_begin) sj = k;
          const ksmi_dst& dj = d.dst[sj];
          int d2 = 0, n2 = n - dj.n_begin;
          if (d.ps_cout > 0) { d2 = n / d.ps_cout; n2 = n - d2 * d.ps_cout; }
          T* q = dst_of(dj, d2, n2);
          float o = v[nf][r];
          if (dj.accumulate) o += ElemTraits<T>::ld(q);
          ElemTraits<T>::st(q, o);
        }
      }
    }
  }
  if (d.stats) {
    // reduce over the 16 pixel lanes (l15) of each k-group with DPP row operations, then over the 4 waves through LDS
    __syncthreads();                                     // (all waves are past their last LDS reads)
    float* red = (float*)smem;                           // [WN groups][4 waves][2][BN]
    const int ws = wn * 4 + wave;
#pragma unroll
    for (int nf = 0; nf < NT; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(ssum[nf][r]), q = row16_sum(ssq[nf][r]);
        const int col = epi_col<NT>(nf, g) + r;
        if (l15 == 0) { red[(ws * 2 + 0) * BN + col] = a; red[(ws * 2 + 1) * BN + col] = q; }
      }
    __syncthreads();
    if (tid < 2 * BN * WN) {
      const int which = tid / (BN * WN), nn = tid - which * (BN * WN);
      const int grp = nn / BN, n = nn - grp * BN;
      const float* rg = red + (size_t)grp * 4 * 2 * BN;
      const float v = rg[(0 * 2 + which) * BN + n] + rg[(1 * 2 + which) * BN + n] + rg[(2 * 2 + which) * BN + n] + rg[(3 * 2 + which) * BN + n];
      const int nbase = n0wg >= 0 ? n0wg : n0;
      const size_t srow = stats_row >= 0 ? (size_t)stats_row : (size_t)blockIdx.x;     // (two tiles per workgroup: igemm2 MP = 2)
      if (nbase + nn < d.Npad) d.stats[(srow * 2 + which) * d.Npad + nbase + nn] = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Lean epilogue of the common case (bf16, NT = 2, ONE destination, channel groups of 8 aligned, no pixel shuffle / strided
// placement / alpha / residual): the launcher proves the preconditions (igemm2.hip, `fast`), so everything the general epilogue
// decides per lane at run time (segment lookup, vector eligibility, scalar tails) is gone -- ~6400 -> ~500 instructions of
// straight-line code; the general one costs ~11k cycles per workgroup, as much as the rest of a K = 32 convolution.
// Lane (g, l15) owns channels n0 + 8g .. 8g+7 of pixel l15 of each 16-pixel row group: one 16-byte access per (lane, mf).
// -------------------------------------------------------------------------------------------------
template <int WN>
__device__ __forceinline__ void igemm_epilogue_fast(const ksmi_conv_desc& d, f32x4 (&acc)[4][2], unsigned char* smem, int tid,
                                                    int wave, int g, int l15, int b, int oy0, int ox0, int n0, int P,
                                                    uint32_t magic_tw, int wn, int n0wg, int stats_row = -1) {
  typedef bf16_t T;
  constexpr int BN = 32;
  const FastDiv dTW(d.TW, magic_tw);
  const int nc = n0 + g * 8;
  const bool nv = nc < d.N;
  const bool has_mask = d.mask_src != nullptr;
  const bool accum = d.dst[0].accumulate != 0;
  const int dC = d.dst[0].C;
  // pixel shuffle (ConvTranspose2d k2 s2 as a 1x1 GEMM over 4*C columns): column group nc -> quadrant dd, channel nn
  int dd = 0, nn = nc;
  if (d.ps_cout > 0) { dd = nc / d.ps_cout; nn = nc - dd * d.ps_cout; }
  T* const obase = (T*)d.dst[0].ptr + d.dst[0].c_off + nn;
  const T* const mbase = (const T*)d.mask_src + nc;
  // pixel offsets and the loads that do not depend on the accumulators go first
  uint32_t opix[4];
  bool ok[4];
  u32x4 mv[4], ov[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    const int p = wave * 64 + mf * 16 + l15;
    const int ly = dTW.div(p), lx = p - ly * d.TW;
    const int oy = oy0 + ly, ox = ox0 + lx;
    ok[mf] = nv && p < P && oy < d.Hout && ox < d.Wout;
    opix[mf] = d.ps_cout > 0 ? (uint32_t)((b * 2 * d.Hout + 2 * oy + (dd >> 1)) * (2 * d.Wout) + 2 * ox + (dd & 1))
                             : (uint32_t)((b * d.Hout + oy) * d.Wout + ox);
    mv[mf] = (u32x4){0u, 0u, 0u, 0u}; ov[mf] = (u32x4){0u, 0u, 0u, 0u};
    if (has_mask && ok[mf]) mv[mf] = *(const u32x4*)(mbase + (size_t)opix[mf] * d.N);
    if (accum && ok[mf]) ov[mf] = *(const u32x4*)(obase + (size_t)opix[mf] * dC);
  }
  float bias[8], mm[8], mr[8], mg[8], mb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias[j] = 0.f; mm[j] = 0.f; mr[j] = 0.f; mg[j] = 0.f; mb[j] = 0.f; }
  auto ld8 = [&](const float* q, float* o, int at) {
    const f32x4 a = *(const f32x4*)(q + at), c = *(const f32x4*)(q + at + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = c[j]; }
  };
  if (d.bias && nv) ld8(d.bias, bias, nn);
  if (has_mask && nv) { ld8(d.m_mean, mm, nc); ld8(d.m_rstd, mr, nc); ld8(d.m_scale, mg, nc); ld8(d.m_shift, mb, nc); }
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = acc[mf][0][r] + bias[r]; v[4 + r] = acc[mf][1][r] + bias[4 + r]; }
    if (has_mask) {
      float m[8];
      vec_unpack<T>(mv[mf], m);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (m[j] - mm[j]) * mr[j];
        if (!(m[j] * mg[j] + mb[j] > 0.f)) v[j] = 0.f;
        const float q = ElemTraits<T>::cvt(v[j]);                    // statistics of the STORED value, round 5
        if (ok[mf]) { ssum[j] += q; ssq[j] += q * xh; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) if (ok[mf]) { const float q = ElemTraits<T>::cvt(v[j]); ssum[j] += q; ssq[j] += q * q; }
    }
    if (accum) {
      float o[8];
      vec_unpack<T>(ov[mf], o);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    if (ok[mf]) *(u32x4*)(obase + (size_t)opix[mf] * dC) = vec_pack<T>(v);
  }
  if (d.stats) {
    __syncthreads();                                     // (all waves are past their last LDS reads)
    float* red = (float*)smem;                           // [WN groups][4 waves][2][BN]
    const int ws = wn * 4 + wave;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = row16_sum(ssum[j]), q = row16_sum(ssq[j]);
      if (l15 == 0) { red[(ws * 2 + 0) * BN + g * 8 + j] = a; red[(ws * 2 + 1) * BN + g * 8 + j] = q; }
    }
    __syncthreads();
    if (tid < 2 * BN * WN) {
      const int which = tid / (BN * WN), nn = tid - which * (BN * WN);
      const int grp = nn / BN, n = nn - grp * BN;
      const float* rg = red + (size_t)grp * 4 * 2 * BN;
      const float v = rg[(0 * 2 + which) * BN + n] + rg[(1 * 2 + which) * BN + n] + rg[(2 * 2 + which) * BN + n] + rg[(3 * 2 + which) * BN + n];
      const size_t srow = stats_row >= 0 ? (size_t)stats_row : (size_t)blockIdx.x;
      if (n0wg + nn < d.Npad) d.stats[(srow * 2 + which) * d.Npad + n0wg + nn] = v;
    }
  }
}
