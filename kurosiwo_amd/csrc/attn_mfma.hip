// MFMA attention for the two attention shapes of the hot path (bf16 activations):
//   FloodViT  (vision_transformer.py:50-66)  : q,k,v = columns of qkv[B*N][3*H*64], N = 197 keys
//   ChangeFormer (changeformer.py:186-208)   : q[B*Nq][C], kv[B*49][2C], head dim 64 or 80, Nq up to 3136
// softmax(q k^T * scale) v per head, flash-style: scores never leave the registers.
//
// Orientation trick.  With D = A.B on v_mfma (lane holds A row / B column l15 = lane & 15 and the k-group g = lane >> 4;
// C/D: column l15, rows g*4 + r) the product S^T = K . Q^T puts 4 consecutive KEYS of one QUERY in a lane's registers,
// which is exactly the B-operand layout (k = g*4 + e) of v_mfma_f32_16x16x16_bf16: P^T feeds O^T = V^T . P^T straight from
// registers, V^T / K^T are read transposed from their row-major LDS images with ds_read_b64_tr_b16.  The key-parallel
// half of the backward pass uses the other orientation (S = Q . K^T) for the same reason.
//
//   forward      : wave = 16 queries, loops over the key tiles              -> out, lse
//   backward (q) : wave = 16 queries: dq, delta = rowsum(dO * O)            -> dq, delta
//   backward (kv): wave = 16 keys, loops over its query slab                -> dK, dV (fp32 partial per query split)
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

struct AttnP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;     // already offset to column 0 of head 0
  bf16_t* o; float* lse;
  const bf16_t* dout; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  float* delta; float* partial;                           // [B][H][Nq] ; [qsplit][B][H][Nk][2][D]
  int q_rs, k_rs, v_rs, o_rs, dq_rs, dk_rs, dv_rs;        // row strides (elements)
  int Nq, Nk, H, B;
  float scale;
  int qsplit, q_per_split;
  uint32_t drop_thr, drop_site; float drop_inv; const uint32_t* rng;   // attn_drop (changeformer.py:160,203): 0 = off
};

__device__ __forceinline__ f32x4 mma16k(f32x4 acc, s16x4 a, s16x4 b) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
  s16x4 r; r[0] = (short)f32_to_bf16(a); r[1] = (short)f32_to_bf16(b); r[2] = (short)f32_to_bf16(c); r[3] = (short)f32_to_bf16(d);
  return r;
}
// transposed 4-row read: rows row0..row0+3 of a row-major LDS image (row stride rs bytes), 16 columns starting at byte
// column cb; lane l15 receives column l15 of the 4 rows
__device__ __forceinline__ s16x4 tr4(const unsigned char* base, int row0, int rs, int cb, int l15) {
  const unsigned a = (unsigned)(uintptr_t)(base + (row0 + (l15 >> 2)) * rs + cb + (l15 & 3) * 8);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a);
}
__device__ __forceinline__ u32x4 ld16(const bf16_t* p, bool ok) {
  return ok ? *(const u32x4*)p : (u32x4){0u, 0u, 0u, 0u};
}
__device__ __forceinline__ float xg_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float xg_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }

template <int D> struct Geo {
  static constexpr int DP = (D + 31) / 32 * 32;      // padded head dim (MFMA k steps of 32)
  static constexpr int KS = DP / 32;
  static constexpr int DT = (D + 15) / 16;           // 16-wide d tiles
  static constexpr int RS = DP * 2 + 16;             // LDS row stride in bytes (16 B pad: spreads the banks)
};

// rows [0, nrows) x DP columns of a [.. ][rs] global matrix into LDS (zero beyond n_valid rows / D columns)
template <int D, int NTHR = 256>
__device__ __forceinline__ void stage_rows(unsigned char* lds, const bf16_t* src, int rs, int nrows, int n_valid, int tid) {
  constexpr int DP = Geo<D>::DP, RS = Geo<D>::RS, VPR = DP / 8;
  for (int i = tid; i < nrows * VPR; i += NTHR) {
    const int row = i / VPR, c8 = i - row * VPR;
    *(u32x4*)(lds + row * RS + c8 * 16) = ld16(src + (size_t)row * rs + c8 * 8, row < n_valid && c8 * 8 < D);
  }
}

// ------------------------------------------------------------------------------------------------ forward
// NW waves per workgroup; the workgroup walks the query tiles blockIdx.x*NW + wave, + gridDim.x*NW, ... (ViT shape: ONE workgroup of
// 8 waves per (b, h) stages K and V once for all 13 query tiles; the 4-wave grid of 64-query workgroups staged them 4 times)
template <int D, int NKT, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_mfma_fwd_kernel(AttnP p) {   // 4 waves: <= 256 VGPRs, two workgroups per CU
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NKT * 16 * G::RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  stage_rows<D, 64 * NW>(Ks, p.k + (size_t)b * p.Nk * p.k_rs + h * D, p.k_rs, NKT * 16, p.Nk, tid);
  stage_rows<D, 64 * NW>(Vs, p.v + (size_t)b * p.Nk * p.v_rs + h * D, p.v_rs, NKT * 16, p.Nk, tid);
  __syncthreads();
  for (int q0 = (blockIdx.x * NW + wave) * 16; q0 < p.Nq; q0 += gridDim.x * NW * 16) {
  const int qrow = q0 + l15;
  const bool qok = qrow < p.Nq;
  const bf16_t* qp = p.q + ((size_t)b * p.Nq + qrow) * p.q_rs + h * D;
  u32x4 Qf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) Qf[ks] = ld16(qp + ks * 32 + g * 8, qok && ks * 32 + g * 8 < D);
  f32x4 S[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const u32x4 a = *(const u32x4*)(Ks + (kt * 16 + l15) * G::RS + (ks * 32 + g * 8) * 2);
      mma16<bf16_t>(acc, a, Qf[ks]);
    }
    S[kt] = acc;
  }
  // softmax over the keys of query l15 (registers: keys kt*16 + g*4 + r; across the 4 k-groups by lane shuffles)
  float m = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool kok = kt * 16 + g * 4 + r < p.Nk;
      S[kt][r] = kok ? S[kt][r] * p.scale : -3.0e38f;
      m = fmaxf(m, S[kt][r]);
    }
  m = xg_max(m);
  float l = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = (kt * 16 + g * 4 + r < p.Nk) ? __expf(S[kt][r] - m) : 0.f;
      S[kt][r] = e;
      l += e;
    }
  l = xg_sum(l);
  if (p.drop_thr) {                                        // attn_drop on the normalised probabilities: the row sum stays undropped
    const uint32_t key = ksmi_rng_key(p.rng, p.drop_site);
    const uint32_t base = (((uint32_t)b * p.H + h) * p.Nq + qrow) * p.Nk;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        S[kt][r] = ksmi_rng_keep(key, base + kt * 16 + g * 4 + r, p.drop_thr) ? S[kt][r] * p.drop_inv : 0.f;
  }
  f32x4 O[G::DT];
#pragma unroll
  for (int dt = 0; dt < G::DT; ++dt) O[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const s16x4 pb = pack4(S[kt][0], S[kt][1], S[kt][2], S[kt][3]);
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) O[dt] = mma16k(O[dt], tr4(Vs, kt * 16 + g * 4, G::RS, dt * 32, l15), pb);
  }
  if (!qok) continue;
  const float inv = 1.f / l;
  bf16_t* op = p.o + ((size_t)b * p.Nq + qrow) * p.o_rs + h * D;
#pragma unroll
  for (int dt = 0; dt < G::DT; ++dt) {
    const int d0 = dt * 16 + g * 4;
    if (d0 < D) {
      uint2 w;
      w.x = (uint32_t)f32_to_bf16(O[dt][0] * inv) | ((uint32_t)f32_to_bf16(O[dt][1] * inv) << 16);
      w.y = (uint32_t)f32_to_bf16(O[dt][2] * inv) | ((uint32_t)f32_to_bf16(O[dt][3] * inv) << 16);
      *(uint2*)(op + d0) = w;
    }
  }
  if (g == 0 && p.lse) p.lse[((size_t)b * p.H + h) * p.Nq + qrow] = m + __logf(l);
  }
}

// ------------------------------------------------------------------------------------------------ backward: dq, delta
// RECOMP: the forward did not save lse (ChangeFormer shape, <= 4 key tiles): recompute it here and publish it for the kv pass
template <int D, int NKT, bool RECOMP>
__global__ __launch_bounds__(256) void attn_mfma_bwd_q_kernel(AttnP p) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + NKT * 16 * G::RS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  stage_rows<D>(Ks, p.k + (size_t)b * p.Nk * p.k_rs + h * D, p.k_rs, NKT * 16, p.Nk, tid);
  stage_rows<D>(Vs, p.v + (size_t)b * p.Nk * p.v_rs + h * D, p.v_rs, NKT * 16, p.Nk, tid);
  __syncthreads();
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= p.Nq) return;
  const int qrow = q0 + l15;
  const bool qok = qrow < p.Nq;
  const bf16_t* qp = p.q + ((size_t)b * p.Nq + qrow) * p.q_rs + h * D;
  const bf16_t* gp = p.dout + ((size_t)b * p.Nq + qrow) * p.o_rs + h * D;
  const bf16_t* op = p.o + ((size_t)b * p.Nq + qrow) * p.o_rs + h * D;
  u32x4 Qf[G::KS], Gf[G::KS];
  float dsum = 0.f;
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    const bool ok = qok && ks * 32 + g * 8 < D;
    Qf[ks] = ld16(qp + ks * 32 + g * 8, ok);
    Gf[ks] = ld16(gp + ks * 32 + g * 8, ok);
    const u32x4 of = ld16(op + ks * 32 + g * 8, ok);
    float a[8], c[8];
    vec_unpack<bf16_t>(Gf[ks], a);
    vec_unpack<bf16_t>(of, c);
#pragma unroll
    for (int j = 0; j < 8; ++j) dsum += a[j] * c[j];
  }
  const float delta = xg_sum(dsum);
  const size_t ridx = ((size_t)b * p.H + h) * p.Nq + qrow;
  if (g == 0 && qok) p.delta[ridx] = delta;
  f32x4 S[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
      mma16<bf16_t>(acc, *(const u32x4*)(Ks + (kt * 16 + l15) * G::RS + (ks * 32 + g * 8) * 2), Qf[ks]);
    S[kt] = acc;
  }
  float lse;
  if constexpr (RECOMP) {
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (kt * 16 + g * 4 + r < p.Nk) m = fmaxf(m, S[kt][r] * p.scale);
    m = xg_max(m);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (kt * 16 + g * 4 + r < p.Nk) l += __expf(S[kt][r] * p.scale - m);
    l = xg_sum(l);
    lse = m + __logf(l);
    if (g == 0 && qok) p.lse[ridx] = lse;
  } else {
    lse = qok ? p.lse[ridx] : 0.f;
  }
  f32x4 dQ[G::DT];
#pragma unroll
  for (int dt = 0; dt < G::DT; ++dt) dQ[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint32_t dkey = p.drop_thr ? ksmi_rng_key(p.rng, p.drop_site) : 0u;
  const uint32_t dbase = (uint32_t)ridx * p.Nk;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    const f32x4 s = S[kt];
    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
      mma16<bf16_t>(dp, *(const u32x4*)(Vs + (kt * 16 + l15) * G::RS + (ks * 32 + g * 8) * 2), Gf[ks]);
    float ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool kok = kt * 16 + g * 4 + r < p.Nk;
      const float pr = (kok && qok) ? __expf(s[r] * p.scale - lse) : 0.f;
      float dpr = dp[r];
      if (p.drop_thr) dpr = ksmi_rng_keep(dkey, dbase + kt * 16 + g * 4 + r, p.drop_thr) ? dpr * p.drop_inv : 0.f;
      ds[r] = pr * (dpr - delta) * p.scale;
    }
    const s16x4 db = pack4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) dQ[dt] = mma16k(dQ[dt], tr4(Ks, kt * 16 + g * 4, G::RS, dt * 32, l15), db);
  }
  if (!qok) return;
  bf16_t* dqp = p.dq + ((size_t)b * p.Nq + qrow) * p.dq_rs + h * D;
#pragma unroll
  for (int dt = 0; dt < G::DT; ++dt) {
    const int d0 = dt * 16 + g * 4;
    if (d0 < D) {
      uint2 w;
      w.x = (uint32_t)f32_to_bf16(dQ[dt][0]) | ((uint32_t)f32_to_bf16(dQ[dt][1]) << 16);
      w.y = (uint32_t)f32_to_bf16(dQ[dt][2]) | ((uint32_t)f32_to_bf16(dQ[dt][3]) << 16);
      *(uint2*)(dqp + d0) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// grid (key chunks of 64 * qsplit, H, B); wave = key tile kchunk*4 + wave; the workgroup walks its query slab 64 rows at a time
template <int D>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kv_kernel(AttnP p, int kchunks) {
  using G = Geo<D>;
  constexpr int QT = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Qs = smem;
  unsigned char* Gs = smem + QT * G::RS;
  float* Ls = (float*)(smem + 2 * QT * G::RS);           // lse[QT], delta[QT]
  float* Ds = Ls + QT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kchunk = blockIdx.x % kchunks, split = blockIdx.x / kchunks;
  const int key = (kchunk * 4 + wave) * 16 + l15;
  const bool kok = key < p.Nk;
  const bf16_t* kp = p.k + ((size_t)b * p.Nk + key) * p.k_rs + h * D;
  const bf16_t* vp = p.v + ((size_t)b * p.Nk + key) * p.v_rs + h * D;
  u32x4 Kf[G::KS], Vf[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) {
    const bool ok = kok && ks * 32 + g * 8 < D;
    Kf[ks] = ld16(kp + ks * 32 + g * 8, ok);
    Vf[ks] = ld16(vp + ks * 32 + g * 8, ok);
  }
  f32x4 dK[G::DT], dV[G::DT];
#pragma unroll
  for (int dt = 0; dt < G::DT; ++dt) { dK[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const int qb = split * p.q_per_split, qe = min(p.Nq, qb + p.q_per_split);
  const uint32_t dkey = p.drop_thr ? ksmi_rng_key(p.rng, p.drop_site) : 0u;
  for (int q0 = qb; q0 < qe; q0 += QT) {
    __syncthreads();
    const int nv = min(QT, qe - q0);
    stage_rows<D>(Qs, p.q + ((size_t)b * p.Nq + q0) * p.q_rs + h * D, p.q_rs, QT, nv, tid);
    stage_rows<D>(Gs, p.dout + ((size_t)b * p.Nq + q0) * p.o_rs + h * D, p.o_rs, QT, nv, tid);
    if (tid < QT) {
      const bool ok = tid < nv;
      const size_t ridx = ((size_t)b * p.H + h) * p.Nq + q0 + tid;
      Ls[tid] = ok ? p.lse[ridx] : 3.0e38f;               // exp(s - 3e38) = 0 for the padded query rows
      Ds[tid] = ok ? p.delta[ridx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < QT / 16; ++t) {
      if (t * 16 >= nv) break;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const int off = (t * 16 + l15) * G::RS + (ks * 32 + g * 8) * 2;
        mma16<bf16_t>(s, *(const u32x4*)(Qs + off), Kf[ks]);          // S[q = g*4+r][key = l15]
        mma16<bf16_t>(dp, *(const u32x4*)(Gs + off), Vf[ks]);         // dP[q][key]
      }
      float pr[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = t * 16 + g * 4 + r;
        pr[r] = kok ? __expf(s[r] * p.scale - Ls[qi]) : 0.f;
        float mk = 1.f;
        if (p.drop_thr)
          mk = ksmi_rng_keep(dkey, ((((uint32_t)b * p.H + h) * p.Nq + q0 + qi) * p.Nk + key), p.drop_thr) ? p.drop_inv : 0.f;
        ds[r] = pr[r] * (dp[r] * mk - Ds[qi]) * p.scale;
        pr[r] *= mk;                                       // dV sees the dropped probabilities
      }
      const s16x4 pb = pack4(pr[0], pr[1], pr[2], pr[3]), db = pack4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        dV[dt] = mma16k(dV[dt], tr4(Gs, t * 16 + g * 4, G::RS, dt * 32, l15), pb);   // dV^T[d][key] += dO^T[d][q] P[q][key]
        dK[dt] = mma16k(dK[dt], tr4(Qs, t * 16 + g * 4, G::RS, dt * 32, l15), db);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
      }
    }
  }
  if (!kok) return;
  if (p.qsplit == 1) {
    bf16_t* dkp = p.dk + ((size_t)b * p.Nk + key) * p.dk_rs + h * D;
    bf16_t* dvp = p.dv + ((size_t)b * p.Nk + key) * p.dv_rs + h * D;
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
      const int d0 = dt * 16 + g * 4;
      if (d0 < D) {
        uint2 w;
        w.x = (uint32_t)f32_to_bf16(dK[dt][0]) | ((uint32_t)f32_to_bf16(dK[dt][1]) << 16);
        w.y = (uint32_t)f32_to_bf16(dK[dt][2]) | ((uint32_t)f32_to_bf16(dK[dt][3]) << 16);
        *(uint2*)(dkp + d0) = w;
        w.x = (uint32_t)f32_to_bf16(dV[dt][0]) | ((uint32_t)f32_to_bf16(dV[dt][1]) << 16);
        w.y = (uint32_t)f32_to_bf16(dV[dt][2]) | ((uint32_t)f32_to_bf16(dV[dt][3]) << 16);
        *(uint2*)(dvp + d0) = w;
      }
    }
  } else {
    float* pp = p.partial + ((((size_t)split * p.B + b) * p.H + h) * p.Nk + key) * 2 * D;
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
      const int d0 = dt * 16 + g * 4;
      if (d0 < D) {
        *(f32x4*)(pp + d0) = dK[dt];
        *(f32x4*)(pp + D + d0) = dV[dt];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, one workgroup per (b, h)
// ViT shape (Nq = Nk <= 208, d = 64): Q, K, V and dO of one head fit the LDS together (4 x 30 KB), so the whole backward of a head is
// one workgroup with no slab loop and no partial sums: phase A (waves over query tiles, S^T orientation) writes dq and delta, phase B
// (waves over key tiles, S orientation) walks all query tiles from LDS and writes dk, dv.  One launch of B*H workgroups (256 for the
// FloodViT batch = one per CU) replaces bwd_q + bwd_kv (+ finish): 71 -> ~25 us per layer.
// NW (round 6): waves per workgroup.  8 = two rounds of query / key tiles per phase at 13 tiles (the second with 5 of 8 waves); 16 = one
// round per phase: the kernel is a chain of LDS / MFMA / exp latencies (3.4 k MFMAs per head = ~3 us of issue against 40-50 us measured),
// so halving the serial rounds matters more than the idle three waves.
template <int D, int NKT, int NW = 8>
__global__ __launch_bounds__(64 * NW) void attn_mfma_bwd_fused_kernel(AttnP p) {
  using G = Geo<D>;
  constexpr int ROWS = NKT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vs = Ks + ROWS * G::RS;
  unsigned char* Qs = Vs + ROWS * G::RS;
  unsigned char* Gs = Qs + ROWS * G::RS;
  float* Ls = (float*)(Gs + ROWS * G::RS);
  float* Ds = Ls + ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.y, h = blockIdx.x;
  stage_rows<D, 64 * NW>(Ks, p.k + (size_t)b * p.Nk * p.k_rs + h * D, p.k_rs, ROWS, p.Nk, tid);
  stage_rows<D, 64 * NW>(Vs, p.v + (size_t)b * p.Nk * p.v_rs + h * D, p.v_rs, ROWS, p.Nk, tid);
  stage_rows<D, 64 * NW>(Qs, p.q + (size_t)b * p.Nq * p.q_rs + h * D, p.q_rs, ROWS, p.Nq, tid);
  stage_rows<D, 64 * NW>(Gs, p.dout + (size_t)b * p.Nq * p.o_rs + h * D, p.o_rs, ROWS, p.Nq, tid);
  for (int i = tid; i < ROWS; i += 64 * NW) Ls[i] = i < p.Nq ? p.lse[((size_t)b * p.H + h) * p.Nq + i] : 3.0e38f;   // exp(s - 3e38) = 0: padded queries
  __syncthreads();
  // ---- phase A: dq and delta, wave = query tile ------------------------------------------------------------------
  for (int qt = wave; qt * 16 < p.Nq; qt += NW) {            // 8 waves: 13 tiles = at most 2 per wave; 16: one
    const int qrow = qt * 16 + l15;
    const bool qok = qrow < p.Nq;
    const bf16_t* op = p.o + ((size_t)b * p.Nq + qrow) * p.o_rs + h * D;
    u32x4 Qf[G::KS], Gf[G::KS];
    float dsum = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int off = qrow * G::RS + (ks * 32 + g * 8) * 2;
      Qf[ks] = *(const u32x4*)(Qs + off);
      Gf[ks] = *(const u32x4*)(Gs + off);
      const u32x4 of = ld16(op + ks * 32 + g * 8, qok && ks * 32 + g * 8 < D);
      float a[8], c[8];
      vec_unpack<bf16_t>(Gf[ks], a);
      vec_unpack<bf16_t>(of, c);
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum += a[j] * c[j];
    }
    const float delta = xg_sum(dsum);
    if (g == 0) Ds[qrow] = qok ? delta : 0.f;
    const float lse = Ls[qrow];
    f32x4 dQ[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) dQ[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const int off = (kt * 16 + l15) * G::RS + (ks * 32 + g * 8) * 2;
        mma16<bf16_t>(s, *(const u32x4*)(Ks + off), Qf[ks]);            // S^T[key = g*4+r][q = l15]
        mma16<bf16_t>(dp, *(const u32x4*)(Vs + off), Gf[ks]);
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool kok = kt * 16 + g * 4 + r < p.Nk;
        const float pr = (kok && qok) ? __expf(s[r] * p.scale - lse) : 0.f;
        ds[r] = pr * (dp[r] - delta) * p.scale;
      }
      const s16x4 db = pack4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) dQ[dt] = mma16k(dQ[dt], tr4(Ks, kt * 16 + g * 4, G::RS, dt * 32, l15), db);
    }
    if (qok) {
      bf16_t* dqp = p.dq + ((size_t)b * p.Nq + qrow) * p.dq_rs + h * D;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        const int d0 = dt * 16 + g * 4;
        if (d0 < D) {
          uint2 w;
          w.x = (uint32_t)f32_to_bf16(dQ[dt][0]) | ((uint32_t)f32_to_bf16(dQ[dt][1]) << 16);
          w.y = (uint32_t)f32_to_bf16(dQ[dt][2]) | ((uint32_t)f32_to_bf16(dQ[dt][3]) << 16);
          *(uint2*)(dqp + d0) = w;
        }
      }
    }
  }
  __syncthreads();                                        // delta of every query is in LDS
  // ---- phase B: dk and dv, wave = key tile -----------------------------------------------------------------------
  for (int kt = wave; kt * 16 < p.Nk; kt += NW) {
    const int key = kt * 16 + l15;
    const bool kok = key < p.Nk;
    u32x4 Kf[G::KS], Vf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int off = key * G::RS + (ks * 32 + g * 8) * 2;
      Kf[ks] = *(const u32x4*)(Ks + off);
      Vf[ks] = *(const u32x4*)(Vs + off);
    }
    f32x4 dK[G::DT], dV[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) { dK[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1                  // (unrolled by 2: neutral, profiles/r06_ab_attn_waves.txt)
    for (int t = 0; t * 16 < p.Nq; ++t) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const int off = (t * 16 + l15) * G::RS + (ks * 32 + g * 8) * 2;
        mma16<bf16_t>(s, *(const u32x4*)(Qs + off), Kf[ks]);          // S[q = g*4+r][key = l15]
        mma16<bf16_t>(dp, *(const u32x4*)(Gs + off), Vf[ks]);
      }
      float pr[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = t * 16 + g * 4 + r;
        pr[r] = kok ? __expf(s[r] * p.scale - Ls[qi]) : 0.f;
        ds[r] = pr[r] * (dp[r] - Ds[qi]) * p.scale;
      }
      const s16x4 pb = pack4(pr[0], pr[1], pr[2], pr[3]), db = pack4(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        dV[dt] = mma16k(dV[dt], tr4(Gs, t * 16 + g * 4, G::RS, dt * 32, l15), pb);
        dK[dt] = mma16k(dK[dt], tr4(Qs, t * 16 + g * 4, G::RS, dt * 32, l15), db);
      }
    }
    if (kok) {
      bf16_t* dkp = p.dk + ((size_t)b * p.Nk + key) * p.dk_rs + h * D;
      bf16_t* dvp = p.dv + ((size_t)b * p.Nk + key) * p.dv_rs + h * D;
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt) {
        const int d0 = dt * 16 + g * 4;
        if (d0 < D) {
          uint2 w;
          w.x = (uint32_t)f32_to_bf16(dK[dt][0]) | ((uint32_t)f32_to_bf16(dK[dt][1]) << 16);
          w.y = (uint32_t)f32_to_bf16(dK[dt][2]) | ((uint32_t)f32_to_bf16(dK[dt][3]) << 16);
          *(uint2*)(dkp + d0) = w;
          w.x = (uint32_t)f32_to_bf16(dV[dt][0]) | ((uint32_t)f32_to_bf16(dV[dt][1]) << 16);
          w.y = (uint32_t)f32_to_bf16(dV[dt][2]) | ((uint32_t)f32_to_bf16(dV[dt][3]) << 16);
          *(uint2*)(dvp + d0) = w;
        }
      }
    }
  }
}

// dk/dv = sum over the query splits of the fp32 partials
template <int D>
__global__ void attn_kv_finish_kernel(AttnP p) {
  const size_t n = (size_t)p.B * p.H * p.Nk * 2 * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < p.qsplit; ++k) s += p.partial[(size_t)k * n + i];
    const int d = i % D; size_t r = i / D;
    const int which = r % 2; r /= 2;
    const int key = r % p.Nk; r /= p.Nk;
    const int h = r % p.H; const int b = r / p.H;
    bf16_t* dst = which ? p.dv + ((size_t)b * p.Nk + key) * p.dv_rs : p.dk + ((size_t)b * p.Nk + key) * p.dk_rs;
    dst[h * D + d] = f32_to_bf16(s);
  }
}

template <typename K>
void set_lds(K kfn, size_t lds) {
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int D, int NKT>
int launch_fwd(const AttnP& p, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 16 * Geo<D>::RS;
  auto kfn = attn_mfma_fwd_kernel<D, NKT>; KSMI_NOTE(kfn);
  set_lds(kfn, lds);
  hipLaunchKernelGGL(kfn, dim3((p.Nq + 63) / 64, p.H, p.B), dim3(256), lds, st, p);
  return ksmi_check_launch("attn_mfma_fwd");
}

template <int D, int NKT, bool RECOMP>
int launch_bwd(AttnP p, hipStream_t st) {
  const size_t lds = 2 * (size_t)NKT * 16 * Geo<D>::RS;
  auto kq = attn_mfma_bwd_q_kernel<D, NKT, RECOMP>;
  set_lds(kq, lds);
  hipLaunchKernelGGL(kq, dim3((p.Nq + 63) / 64, p.H, p.B), dim3(256), lds, st, p);
  int rc = ksmi_check_launch("attn_mfma_bwd_q");
  if (rc) return rc;
  const int kchunks = (p.Nk + 63) / 64;
  const size_t lds2 = 2 * (size_t)64 * Geo<D>::RS + 2 * 64 * sizeof(float);
  hipLaunchKernelGGL(attn_mfma_bwd_kv_kernel<D>, dim3(kchunks * p.qsplit, p.H, p.B), dim3(256), lds2, st, p, kchunks);
  rc = ksmi_check_launch("attn_mfma_bwd_kv");
  if (rc || p.qsplit == 1) return rc;
  const size_t n = (size_t)p.B * p.H * p.Nk * 2 * D;
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(attn_kv_finish_kernel<D>, dim3(blocks), dim3(256), 0, st, p);
  return ksmi_check_launch("attn_kv_finish");
}

}  // namespace

// query splits of the key-parallel backward so that the grid holds >= ~512 workgroups
int ksmi_attn_mfma_qsplit(int B, int Nq, int Nk, int H) {
  const int base = ((Nk + 63) / 64) * H * B;
  int want = (512 + base - 1) / base;
  const int maxs = (Nq + 63) / 64;
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 32) want = 32;
  return want;
}

size_t ksmi_attn_mfma_workspace(int B, int Nq, int Nk, int H, int D) {
  const int qs = ksmi_attn_mfma_qsplit(B, Nq, Nk, H);
  return 2 * ((((size_t)B * H * Nq + 63) & ~(size_t)63) * sizeof(float)) + (qs > 1 ? (size_t)qs * B * H * Nk * 2 * D * sizeof(float) : 0) + 256;
}

static void fill_split(AttnP& p, void* workspace) {
  p.qsplit = ksmi_attn_mfma_qsplit(p.B, p.Nq, p.Nk, p.H);
  p.q_per_split = (((p.Nq + p.qsplit - 1) / p.qsplit) + 63) & ~63;
  p.qsplit = (p.Nq + p.q_per_split - 1) / p.q_per_split;
  const size_t nrow = ((size_t)p.B * p.H * p.Nq + 63) & ~(size_t)63;
  p.delta = (float*)workspace;
  if (!p.lse) p.lse = p.delta + nrow;                     // recomputed by the q pass
  p.partial = p.delta + 2 * nrow;
}

// FloodViT shape: qkv [B*N][3*H*64]
int ksmi_attn_mfma_vit(int backward, const void* qkv, void* out, float* lse, const void* dout, void* dqkv, void* workspace, int B, int N,
                       int H, float scale, void* stream) {
  if (N > 208) return ksmi_fail(KSMI_E_UNSUPPORTED, "attention (MFMA): at most 208 keys");
  const int C3 = 3 * H * 64;
  AttnP p = {};
  p.q = (const bf16_t*)qkv; p.k = p.q + H * 64; p.v = p.q + 2 * H * 64;
  p.o = (bf16_t*)out; p.lse = lse;
  p.q_rs = p.k_rs = p.v_rs = C3; p.o_rs = H * 64;
  p.Nq = p.Nk = N; p.H = H; p.B = B; p.scale = scale;
  if (!backward) {
    // one 8-wave workgroup per head measured the same as the 64-query workgroups (FloodViT 1011 vs 1016 tiles/s): kept as a switch
    // One 8-wave workgroup per head (K and V staged once for the 13 query tiles) against four 4-wave workgroups of 64 queries that stage
    // them four times.  Round 2 measured the two the same inside the step; re-measured in round 6 (same box, FloodViT step): 27.7 -> 18.6 us
    // per layer, 1235 -> 1260 tiles/s (profiles/r06_ab_attn_waves.txt).  A 16-wave instance (one round of tiles) spills at 128 VGPRs: 49 us.
    // KSMI_ATTN_FWD_ONE_WG=0: the 64-query workgroups.
    static const int one_wg = ksmi_knob_int("KSMI_ATTN_FWD_ONE_WG", 1);
    if (!one_wg) return launch_fwd<64, 13>(p, (hipStream_t)stream);
    const size_t lds = 2 * (size_t)13 * 16 * Geo<64>::RS;
    auto kfn = attn_mfma_fwd_kernel<64, 13, 8>; KSMI_NOTE(kfn);
    set_lds(kfn, lds);
    hipLaunchKernelGGL(kfn, dim3(1, H, B), dim3(512), lds, (hipStream_t)stream, p);
    return ksmi_check_launch("attn_mfma_fwd");
  }
  p.dout = (const bf16_t*)dout;
  p.dq = (bf16_t*)dqkv; p.dk = p.dq + H * 64; p.dv = p.dq + 2 * H * 64;
  p.dq_rs = p.dk_rs = p.dv_rs = C3;
  static const bool split = ksmi_knob_is_set("KSMI_ATTN_SPLIT");      // A/B: the two-kernel backward
  if (!split && lse) {
    constexpr size_t lds = 4 * (size_t)13 * 16 * Geo<64>::RS + 2 * 13 * 16 * sizeof(float);
    // 16 waves = one round of tiles per phase: 39.8 -> 31.1 us per layer in the FloodViT step, 1217 -> 1250 tiles/s same box
    // (profiles/r06_ab_attn_waves.txt); KSMI_ATTN_BWD_NW=8: the round-2 form
    static const int nw = ksmi_knob_int("KSMI_ATTN_BWD_NW", 16);
    if (nw == 16) {
      auto kfn = attn_mfma_bwd_fused_kernel<64, 13, 16>; KSMI_NOTE(kfn);
      set_lds(kfn, lds);
      hipLaunchKernelGGL(kfn, dim3(H, B), dim3(1024), lds, (hipStream_t)stream, p);
      return ksmi_check_launch("attn_mfma_bwd_fused");
    }
    auto kfn = attn_mfma_bwd_fused_kernel<64, 13>; KSMI_NOTE(kfn);
    set_lds(kfn, lds);
    hipLaunchKernelGGL(kfn, dim3(H, B), dim3(512), lds, (hipStream_t)stream, p);
    return ksmi_check_launch("attn_mfma_bwd_fused");
  }
  fill_split(p, workspace);
  return launch_bwd<64, 13, false>(p, (hipStream_t)stream);
}

// ChangeFormer shape: q [B*Nq][C], kv [B*Nk][2C] with Nk <= 64, head dim 64 or 80
int ksmi_attn_mfma_sr(int backward, const void* q, const void* kv, void* out, float* lse, const void* dout, void* dq, void* dkv,
                      void* workspace, int B, int Nq, int Nk, int H, int C, float scale, uint32_t drop_thr, float drop_inv,
                      uint32_t drop_site, const uint32_t* rng, void* stream) {
  if (Nk > 64) return ksmi_fail(KSMI_E_UNSUPPORTED, "sr attention (MFMA): at most 64 keys");
  const int D = C / H;
  AttnP p = {};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)kv; p.v = p.k + C;
  p.o = (bf16_t*)out; p.lse = lse;                        // lse may be null: the backward recomputes it
  p.q_rs = C; p.k_rs = p.v_rs = 2 * C; p.o_rs = C;
  p.Nq = Nq; p.Nk = Nk; p.H = H; p.B = B; p.scale = scale;
  p.drop_thr = drop_thr; p.drop_inv = drop_inv; p.drop_site = drop_site; p.rng = rng;
  if (backward) {
    p.dout = (const bf16_t*)dout;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dkv; p.dv = p.dk + C;
    p.dq_rs = C; p.dk_rs = p.dv_rs = 2 * C;
    fill_split(p, workspace);
  }
  if (D == 64) return backward ? launch_bwd<64, 4, true>(p, (hipStream_t)stream) : launch_fwd<64, 4>(p, (hipStream_t)stream);
  if (D == 80) return backward ? launch_bwd<80, 4, true>(p, (hipStream_t)stream) : launch_fwd<80, 4>(p, (hipStream_t)stream);
  return ksmi_fail(KSMI_E_UNSUPPORTED, "sr attention (MFMA): head dim must be 64 or 80");
}
