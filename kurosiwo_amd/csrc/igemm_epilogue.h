// Shared epilogue of the implicit-GEMM kernels (see igemm.hip / igemm2.hip).
#pragma once
#include "common.h"
#include "../../include/ksmi.h"

// LDS swizzle of the 16-byte k-group slot inside a 64-byte row (conflict-free
// ds_read_b128 for 16 consecutive rows; derivation in DESIGN.md §LDS).
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }

template <typename T, int NT>
__device__ __forceinline__ void igemm_epilogue(const ksmi_conv_desc& d, f32x4 (&acc)[4][NT], unsigned char* smem, int tid,
                                               int wave, int g, int l15, int b, int oy0, int ox0, int n0, int P) {
  constexpr int BN = NT * 16;
  constexpr int VEC = ElemTraits<T>::kVec;
  // ---- epilogue ---------------------------------------------------------------------------------
  // phase 1: accumulators (+bias) -> LDS tile [P][BN] in T (C layout: col n = l15, row = g*4 + r);
  //          plain BatchNorm statistics (sum, sumsq) straight from the fp32 registers.
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
      const int ly = p / d.TW, lx = p - ly * d.TW;
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
        if (rvalid[mf][r] && nvalid) { s_sum[nf] += v; s_sq[nf] += v * v; }
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
      const int ly = p / d.TW, lx = p - ly * d.TW;
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
        dp = (T*)ds.ptr + opix * ds.C + ds.c_off + (nq - ds.n_begin);
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
            q = (T*)dj.ptr + opix * dj.C + dj.c_off + (n - dj.n_begin);
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
