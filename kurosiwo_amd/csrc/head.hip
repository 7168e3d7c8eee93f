// ECAM head (channel attention + fused 1x1 conv_final), fused softmax-CE + Dice loss,
// argmax -> confusion matrix.  HBM-bound: every activation is read once per pass.
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

constexpr int kThreads = 256;
constexpr int kPoolSplit = 64;       // most pixel splits per image of the global pools (workspace size); pool_split() picks the count
constexpr int kBwdSplit = 64;        // pixel splits of the phase-A reduction (the finish kernel is one block per sample)

// ------------------------------------------------------------------------------------------------
// ECAM global avg+max pool over H*W for cat(x0_1..x0_4) [4C] and for their sum [C]
//   workspace: psum[B][S][5C], pmax[B][S][5C], pidx[B][S][5C] (int)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ecam_pool_kernel(const T* x0, const T* x1, const T* x2, const T* x3,
                                                        float* psum, float* pmax, int* pidx, int HW, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int CV = C / VEC;
  const int npl = kThreads / CV;
  const int b = blockIdx.y, sp = blockIdx.x, S = gridDim.x;
  const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
  const bool active = (int)threadIdx.x < npl * CV;
  const int per = (HW + S - 1) / S;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const T* xs[4] = {x0, x1, x2, x3};
  float s[5][VEC], m[5][VEC];
  int mi[5][VEC];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s[k][j] = 0.f; m[k][j] = -INFINITY; mi[k][j] = 0x7fffffff; }
  if (active)
    for (int p = p0 + pl; p < p1; p += npl) {
      float it[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) it[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v[VEC];
        vec_unpack<T>(*(const u32x4*)(xs[k] + ((int64_t)b * HW + p) * C + cv * VEC), v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          s[k][j] += v[j];
          if (v[j] > m[k][j]) { m[k][j] = v[j]; mi[k][j] = p; }
          it[j] += v[j];               // torch.sum(torch.stack(...)) order: ((x1+x2)+x3)+x4
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float q = ElemTraits<T>::cvt(it[j]);   // the reference materialises `intra` in the activation dtype
        s[4][j] += q;
        if (q > m[4][j]) { m[4][j] = q; mi[4][j] = p; }
      }
    }
  // reduce over the pixel lanes.  Lanes with the same channel vector are CV apart: when CV is a power of two that divides 64 the
  // 64 / CV pixel lanes of a wave combine by xor-shuffles in registers and only the 4 wave results go through LDS (the old path
  // walked 64 LDS entries with CV threads, 40 times: it cost more than streaming the four 103 MB tensors)
  if ((CV & (CV - 1)) == 0 && CV <= 64 && C <= 128) {
    __shared__ float ws[4][5 * 128], wm[4][5 * 128];
    __shared__ int wi[4][5 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float a = s[k][j], mx = m[k][j];
        int ix = mi[k][j];
        for (int o = CV; o < 64; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          const float m2 = __shfl_xor(mx, o, 64);
          const int i2 = __shfl_xor(ix, o, 64);
          if (m2 > mx || (m2 == mx && i2 < ix)) { mx = m2; ix = i2; }
        }
        if (lane < CV) {
          const int e = (k * VEC + j) * CV + lane;
          ws[wave][e] = a; wm[wave][e] = mx; wi[wave][e] = ix;
        }
      }
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * VEC * CV; e += kThreads) {
      float a = 0.f, mx = -INFINITY; int ix = 0x7fffffff;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += ws[w][e];
        if (wm[w][e] > mx || (wm[w][e] == mx && wi[w][e] < ix)) { mx = wm[w][e]; ix = wi[w][e]; }
      }
      const int kj = e / CV, cvv = e - kj * CV;
      const int k = kj / VEC, j = kj - k * VEC;
      const int c = k * C + cvv * VEC + j;
      const size_t o = ((size_t)b * S + sp) * 5 * C + c;
      psum[o] = a; pmax[o] = mx; pidx[o] = ix;
    }
    return;
  }
  __shared__ float rs[kThreads], rm[kThreads];
  __shared__ int ri[kThreads];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      __syncthreads();
      rs[threadIdx.x] = active ? s[k][j] : 0.f;
      rm[threadIdx.x] = active ? m[k][j] : -INFINITY;
      ri[threadIdx.x] = active ? mi[k][j] : 0x7fffffff;
      __syncthreads();
      if ((int)threadIdx.x < CV) {
        float a = 0.f, mx = -INFINITY; int ix = 0x7fffffff;
        for (int q = 0; q < npl; ++q) {
          const int t = q * CV + threadIdx.x;
          a += rs[t];
          if (rm[t] > mx || (rm[t] == mx && ri[t] < ix)) { mx = rm[t]; ix = ri[t]; }
        }
        const int c = k * C + threadIdx.x * VEC + j;
        const size_t o = ((size_t)b * S + sp) * 5 * C + c;
        psum[o] = a; pmax[o] = mx; pidx[o] = ix;
      }
    }
}

__global__ void ecam_pool_final_kernel(const float* psum, const float* pmax, const int* pidx, float* avg, float* mx,
                                       int* argmax, int B, int S, int C5, int HW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C5) return;
  const int b = i / C5, c = i - b * C5;
  double a = 0.0; float m = -INFINITY; int ix = 0x7fffffff;
  for (int sp = 0; sp < S; ++sp) {
    const size_t o = ((size_t)b * S + sp) * C5 + c;
    a += (double)psum[o];
    if (pmax[o] > m || (pmax[o] == m && pidx[o] < ix)) { m = pmax[o]; ix = pidx[o]; }
  }
  avg[i] = (float)(a / (double)HW); mx[i] = m; argmax[i] = ix;
}

// tiny MLPs: one block per image.  hidden layout per image: [avg|max][h_ca (4C/16) + h_ca1 (C/4)]
__global__ void ecam_mlp_kernel(const float* avg, const float* mx, const float* w1, const float* w2, const float* v1,
                                const float* v2, float* ca, float* ca1, float* hidden, int C) {
  const int b = blockIdx.x, C4 = 4 * C, H1 = C4 / 16, H2 = C / 4, HT = H1 + H2;
  extern __shared__ float sm[];
  float* h = sm;   // [2][HT]
  for (int i = threadIdx.x; i < 2 * HT; i += blockDim.x) {
    const int which = i / HT, j = i - which * HT;
    const float* in = (which == 0 ? avg : mx) + (size_t)b * 5 * C;
    float a = 0.f;
    if (j < H1) { for (int c = 0; c < C4; ++c) a += w1[j * C4 + c] * in[c]; }
    else { const int jj = j - H1; for (int c = 0; c < C; ++c) a += v1[jj * C + c] * in[C4 + c]; }
    a = fmaxf(a, 0.f);
    h[i] = a; hidden[(size_t)b * 2 * HT + i] = a;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 5 * C; c += blockDim.x) {
    float a = 0.f, m = 0.f;
    if (c < C4) {
      for (int j = 0; j < H1; ++j) { a += w2[c * H1 + j] * h[j]; m += w2[c * H1 + j] * h[HT + j]; }
      ca[(size_t)b * C4 + c] = 1.f / (1.f + expf(-(a + m)));
    } else {
      const int cc = c - C4;
      for (int j = 0; j < H2; ++j) { a += v2[cc * H2 + j] * h[H1 + j]; m += v2[cc * H2 + j] * h[HT + H1 + j]; }
      ca1[(size_t)b * C + cc] = 1.f / (1.f + expf(-(a + m)));
    }
  }
}

// logits[b,k,p] = beff[k] + sum_c weff[k][c] * x[c] ; weff = Wf*ca, beff = bias + sum_c weff*ca1[c%C]
template <typename T, int NCLS>
__global__ __launch_bounds__(256) void ecam_final_fwd_kernel(const T* x0, const T* x1, const T* x2, const T* x3,
                                                             const float* ca, const float* ca1, const float* wf,
                                                             const float* bias, float* logits, int HW, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  extern __shared__ float sm[];
  const int b = blockIdx.y, C4 = 4 * C;
  float* weff = sm;                 // [NCLS][4C]
  float* beff = sm + NCLS * C4;     // [NCLS]
  for (int i = threadIdx.x; i < NCLS * C4; i += blockDim.x) weff[i] = wf[i] * ca[(size_t)b * C4 + (i % C4)];
  __syncthreads();
  if (threadIdx.x < NCLS) {
    float a = bias[threadIdx.x];
    for (int c = 0; c < C4; ++c) a += weff[threadIdx.x * C4 + c] * ca1[(size_t)b * C + (c % C)];
    beff[threadIdx.x] = a;
  }
  __syncthreads();
  const T* xs[4] = {x0, x1, x2, x3};
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float acc[NCLS];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) acc[k] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const T* row = xs[s] + ((int64_t)b * HW + p) * C;
      for (int c0 = 0; c0 < C; c0 += VEC) {
        float v[VEC];
        vec_unpack<T>(*(const u32x4*)(row + c0), v);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
#pragma unroll
          for (int k = 0; k < NCLS; ++k) acc[k] += weff[k * C4 + s * C + c0 + j] * v[j];
      }
    }
#pragma unroll
    for (int k = 0; k < NCLS; ++k) logits[((int64_t)b * NCLS + k) * HW + p] = acc[k] + beff[k];
  }
}

// backward phase A: G[b][k][c] = sum_p dl[b,k,p]*x[b,p,c] (c over 4C), DL[b][k] = sum_p dl[b,k,p]
//   partial workspace: pg[B][S][NCLS][4C], pd[B][S][NCLS]
template <typename T, int NCLS>
__global__ __launch_bounds__(256) void ecam_bwd_reduce_kernel(const T* x0, const T* x1, const T* x2, const T* x3,
                                                              const float* dl, float* pg, float* pd, int HW, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  const int C4 = 4 * C, CV4 = C4 / VEC;            // channel-vectors over the concatenation
  const int npl = kThreads / CV4;
  const int b = blockIdx.y, sp = blockIdx.x, S = gridDim.x;
  const int cv = threadIdx.x % CV4, pl = threadIdx.x / CV4;
  const bool active = (int)threadIdx.x < npl * CV4;
  const int per = (HW + S - 1) / S;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const T* xs[4] = {x0, x1, x2, x3};
  const int src = (cv * VEC) / C, coff = cv * VEC - src * C;
  float g[NCLS][VEC], dsum[NCLS];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) { dsum[k] = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) g[k][j] = 0.f; }
  if (active)
    for (int p = p0 + pl; p < p1; p += npl) {
      float v[VEC];
      vec_unpack<T>(*(const u32x4*)(xs[src] + ((int64_t)b * HW + p) * C + coff), v);
#pragma unroll
      for (int k = 0; k < NCLS; ++k) {
        const float d = dl[((int64_t)b * NCLS + k) * HW + p];
        dsum[k] += d;
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[k][j] += d * v[j];
      }
    }
  // lanes with the same channel vector are CV4 apart: when CV4 is a power of two dividing 64 the 64 / CV4 pixel lanes of a wave
  // combine by xor-shuffles and only the 4 wave results meet in LDS (one barrier instead of 2 x NCLS x (VEC + 1), as in ecam_pool_kernel)
  if ((CV4 & (CV4 - 1)) == 0 && CV4 <= 32) {
    __shared__ float wsum[4][NCLS * VEC * 32 + NCLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float a = g[k][j];
        for (int o = CV4; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        if (lane < CV4) wsum[wave][(k * VEC + j) * CV4 + lane] = a;
      }
      float a = cv == 0 ? dsum[k] : 0.f;
      for (int o = CV4; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0) wsum[wave][NCLS * VEC * CV4 + k] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NCLS * VEC * CV4 + NCLS; e += kThreads) {
      const float a = (wsum[0][e] + wsum[1][e]) + (wsum[2][e] + wsum[3][e]);
      if (e < NCLS * VEC * CV4) {
        const int kj = e / CV4, cvv = e - kj * CV4;
        const int k = kj / VEC, j = kj - k * VEC;
        pg[(((size_t)b * S + sp) * NCLS + k) * C4 + cvv * VEC + j] = a;
      } else {
        pd[((size_t)b * S + sp) * NCLS + (e - NCLS * VEC * CV4)] = a;
      }
    }
    return;
  }
  __shared__ float red[kThreads];
#pragma unroll
  for (int k = 0; k < NCLS; ++k) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      __syncthreads();
      red[threadIdx.x] = active ? g[k][j] : 0.f;
      __syncthreads();
      if ((int)threadIdx.x < CV4) {
        float a = 0.f;
        for (int q = 0; q < npl; ++q) a += red[q * CV4 + threadIdx.x];
        pg[(((size_t)b * S + sp) * NCLS + k) * C4 + threadIdx.x * VEC + j] = a;
      }
    }
    __syncthreads();
    red[threadIdx.x] = (active && cv == 0) ? dsum[k] : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      float a = 0.f;
      for (int q = 0; q < npl; ++q) a += red[q * CV4];
      pd[((size_t)b * S + sp) * NCLS + k] = a;
    }
  }
}

// finish phase A, block = sample b: G[b], DL[b] (fp64 over the splits) -> gbuf / dlbuf, dca[b], dca1[b]
template <int NCLS>
__global__ __launch_bounds__(256) void ecam_bwd_finish_kernel(const float* pg, const float* pd, const float* ca, const float* ca1, const float* wf,
                                                              float* dca, float* dca1, float* gbuf, float* dlbuf, int S, int C) {
  extern __shared__ float sm[];
  const int C4 = 4 * C, b = blockIdx.x;
  float* gs = sm;                      // [NCLS][4C]
  float* dls = sm + NCLS * C4;         // [NCLS]
  for (int i = threadIdx.x; i < NCLS * C4; i += blockDim.x) {
    const int c = i % C4, k = i / C4;
    double a = 0.0;
    for (int sp = 0; sp < S; ++sp) a += (double)pg[(((size_t)b * S + sp) * NCLS + k) * C4 + c];
    gs[i] = (float)a;
    gbuf[(size_t)b * NCLS * C4 + i] = (float)a;
  }
  for (int k = threadIdx.x; k < NCLS; k += blockDim.x) {
    double a = 0.0;
    for (int sp = 0; sp < S; ++sp) a += (double)pd[((size_t)b * S + sp) * NCLS + k];
    dls[k] = (float)a;
    dlbuf[b * NCLS + k] = (float)a;
  }
  __syncthreads();
  // dca[b][c] = sum_k Wf[k][c] * (G[b][k][c] + ca1[b][c%C]*DL[b][k])
  for (int c = threadIdx.x; c < C4; c += blockDim.x) {
    float a = 0.f;
    for (int k = 0; k < NCLS; ++k) a += wf[k * C4 + c] * (gs[k * C4 + c] + ca1[b * C + (c % C)] * dls[k]);
    dca[(size_t)b * C4 + c] = a;
  }
  // dca1[b][cc] = sum_j ca[b][cc+jC] * sum_k Wf[k][cc+jC]*DL[b][k]
  for (int cc = threadIdx.x; cc < C; cc += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < 4; ++j) {
      float s1 = 0.f;
      for (int k = 0; k < NCLS; ++k) s1 += wf[k * C4 + cc + j * C] * dls[k];
      a += ca[(size_t)b * C4 + cc + j * C] * s1;
    }
    dca1[(size_t)b * C + cc] = a;
  }
}

// ... and across the samples: dWf[k][c] = sum_b ca[b][c] * (G[b][k][c] + ca1[b][c%C]*DL[b][k]) (=), dbias[k] = sum_b DL[b][k] (=)
template <int NCLS>
__global__ __launch_bounds__(256) void ecam_bwd_wf_kernel(const float* gbuf, const float* dlbuf, const float* ca, const float* ca1, float* dwf,
                                                          float* dbias, int B, int C) {
  const int C4 = 4 * C;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < NCLS * C4) {
    const int c = i % C4, k = i / C4;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += ca[(size_t)b * C4 + c] * (gbuf[((size_t)b * NCLS + k) * C4 + c] + ca1[b * C + (c % C)] * dlbuf[b * NCLS + k]);
    dwf[i] = a;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < NCLS) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dlbuf[b * NCLS + threadIdx.x];
    dbias[threadIdx.x] = a;
  }
}

// MLP backward, block = sample b: sigmoid' -> fc2 -> relu' -> fc1 (pre-sigmoid / pre-ReLU gradients also to `scratch` for the weights)
__global__ __launch_bounds__(256) void ecam_mlp_bwd_kernel(const float* hidden, const float* ca, const float* ca1, const float* dca,
                                                           const float* dca1, const float* w1, const float* w2, const float* v1, const float* v2,
                                                           float* davg, float* dmax, float* scratch /* [B][5C] ds + [B][2][HT] dh */, int B, int C) {
  extern __shared__ float sm[];
  const int C4 = 4 * C, H1 = C4 / 16, H2 = C / 4, HT = H1 + H2, C5 = 5 * C, b = blockIdx.x;
  float* ds = sm;                      // [5C] pre-sigmoid grad
  float* dh = sm + C5;                 // [2][HT] grad wrt relu input
  float* gds = scratch + (size_t)b * C5;
  float* gdh = scratch + (size_t)B * C5 + (size_t)b * 2 * HT;
  for (int c = threadIdx.x; c < C5; c += blockDim.x) {
    const float s = c < C4 ? ca[(size_t)b * C4 + c] : ca1[(size_t)b * C + c - C4];
    const float d = c < C4 ? dca[(size_t)b * C4 + c] : dca1[(size_t)b * C + c - C4];
    const float v = d * s * (1.f - s);
    ds[c] = v; gds[c] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * HT; i += blockDim.x) {
    const int j = i % HT, which = i / HT;
    const float h = hidden[(size_t)b * 2 * HT + which * HT + j];
    float a = 0.f;
    if (j < H1) { for (int c = 0; c < C4; ++c) a += w2[c * H1 + j] * ds[c]; }
    else { const int jj = j - H1; for (int c = 0; c < C; ++c) a += v2[c * H2 + jj] * ds[C4 + c]; }
    const float v = h > 0.f ? a : 0.f;
    dh[i] = v; gdh[i] = v;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C5; c += blockDim.x) {
    float a = 0.f, m = 0.f;
    if (c < C4) { for (int j = 0; j < H1; ++j) { a += w1[j * C4 + c] * dh[j]; m += w1[j * C4 + c] * dh[HT + j]; } }
    else { const int cc = c - C4; for (int j = 0; j < H2; ++j) { a += v1[j * C + cc] * dh[H1 + j]; m += v1[j * C + cc] * dh[HT + H1 + j]; } }
    davg[(size_t)b * C5 + c] = a; dmax[(size_t)b * C5 + c] = m;
  }
}

// weight grads across the samples (=): fc2: g[c][j] = sum_b ds[b][c]*(h_avg+h_max)[b][j] ; fc1: g[j][c] = sum_b dh_avg*avg + dh_max*max
__global__ __launch_bounds__(256) void ecam_mlp_wgrad_kernel(const float* avg, const float* mx, const float* hidden, const float* scratch,
                                                             float* gw1, float* gw2, float* gv1, float* gv2, int B, int C) {
  const int C4 = 4 * C, H1 = C4 / 16, H2 = C / 4, HT = H1 + H2, C5 = 5 * C;
  const float* ds = scratch;
  const float* dh = scratch + (size_t)B * C5;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C4 * H1) {
    const int j = i % H1, c = i / H1;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += ds[b * C5 + c] * (hidden[(size_t)b * 2 * HT + j] + hidden[(size_t)b * 2 * HT + HT + j]);
    gw2[i] = a;
    return;
  }
  i -= C4 * H1;
  if (i < C * H2) {
    const int j = i % H2, c = i / H2;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += ds[b * C5 + C4 + c] * (hidden[(size_t)b * 2 * HT + H1 + j] + hidden[(size_t)b * 2 * HT + HT + H1 + j]);
    gv2[i] = a;
    return;
  }
  i -= C * H2;
  if (i < H1 * C4) {
    const int c = i % C4, j = i / C4;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dh[(b * 2 + 0) * HT + j] * avg[b * C5 + c] + dh[(b * 2 + 1) * HT + j] * mx[b * C5 + c];
    gw1[i] = a;
    return;
  }
  i -= H1 * C4;
  if (i < H2 * C) {
    const int c = i % C, j = i / C;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dh[(b * 2 + 0) * HT + H1 + j] * avg[b * C5 + C4 + c] + dh[(b * 2 + 1) * HT + H1 + j] * mx[b * C5 + C4 + c];
    gv1[i] = a;
  }
}

// phase B: dx_s[b,p,cc] = ca[b,sC+cc]*sum_k Wf[k][sC+cc]*dl[b,k,p] + (davg[b,sC+cc] + davg[b,4C+cc])/HW
template <typename T, int NCLS>
__global__ __launch_bounds__(256) void ecam_bwd_dx_kernel(T* d0, T* d1, T* d2, T* d3, const float* dl, const float* ca,
                                                          const float* wf, const float* davg, int HW, int C) {
  constexpr int VEC = ElemTraits<T>::kVec;
  extern __shared__ float sm[];
  const int b = blockIdx.y, C4 = 4 * C;
  float* weff = sm;                 // [NCLS][4C]
  float* cst = sm + NCLS * C4;      // [4C]
  const float inv = 1.f / (float)HW;
  for (int i = threadIdx.x; i < NCLS * C4; i += blockDim.x) weff[i] = wf[i] * ca[(size_t)b * C4 + (i % C4)];
  for (int i = threadIdx.x; i < C4; i += blockDim.x) cst[i] = (davg[(size_t)b * 5 * C + i] + davg[(size_t)b * 5 * C + C4 + (i % C)]) * inv;
  __syncthreads();
  // thread = (pixel lane, source, channel vector): its 16-byte store lands next to its neighbours' (one source row = C contiguous
  // channels), and its VEC x (1 + NCLS) coefficients stay in registers across the pixel walk
  const int CV = C / VEC, tpp = 4 * CV;              // threads per pixel
  const int ppb = blockDim.x / tpp;                  // pixels per trip (launcher: tpp <= blockDim.x)
  const int q = threadIdx.x % tpp, pl = threadIdx.x / tpp;
  const int s = q / CV, c0 = (q - s * CV) * VEC;
  if (pl >= ppb) return;
  float w[NCLS][VEC], k0[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    k0[j] = cst[s * C + c0 + j];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) w[k][j] = weff[k * C4 + s * C + c0 + j];
  }
  T* dst = (s == 0 ? d0 : s == 1 ? d1 : s == 2 ? d2 : d3) + (int64_t)b * HW * C + c0;
  for (int p = blockIdx.x * ppb + pl; p < HW; p += gridDim.x * ppb) {
    float d[NCLS], v[VEC];
#pragma unroll
    for (int k = 0; k < NCLS; ++k) d[k] = dl[((int64_t)b * NCLS + k) * HW + p];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float a = k0[j];
#pragma unroll
      for (int k = 0; k < NCLS; ++k) a += w[k][j] * d[k];
      v[j] = a;
    }
    *(u32x4*)(dst + (int64_t)p * C) = vec_pack<T>(v);
  }
}

// max-pool scatter: one thread per (b, cc)
template <typename T>
__global__ void ecam_bwd_max_scatter_kernel(T* d0, T* d1, T* d2, T* d3, const float* dmax, const int* argmax, int B, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, cc = i - b * C;
  T* ds[4] = {d0, d1, d2, d3};
  for (int s = 0; s < 4; ++s) {
    const int c = s * C + cc;
    T* q = ds[s] + ((int64_t)b * HW + argmax[b * 5 * C + c]) * C + cc;
    ElemTraits<T>::st(q, ElemTraits<T>::ld(q) + dmax[b * 5 * C + c]);
  }
  const int pi = argmax[b * 5 * C + 4 * C + cc];
  const float g = dmax[b * 5 * C + 4 * C + cc];
  for (int s = 0; s < 4; ++s) {
    T* q = ds[s] + ((int64_t)b * HW + pi) * C + cc;
    ElemTraits<T>::st(q, ElemTraits<T>::ld(q) + g);
  }
}

// ------------------------------------------------------------------------------------------------
// fused softmax-CE + Dice (3 classes).  workspace: part[B][S][4] then coef[1 + 2B]
// ------------------------------------------------------------------------------------------------
constexpr int kLossSplit = 32;
struct Px3 { float p0, p1, p2; };
__device__ __forceinline__ Px3 softmax3(float a, float b, float c, float& lse) {
  const float m = fmaxf(a, fmaxf(b, c));
  const float e0 = expf(a - m), e1 = expf(b - m), e2 = expf(c - m);
  const float s = e0 + e1 + e2;
  lse = m + logf(s);
  const float r = 1.f / s;
  return {e0 * r, e1 * r, e2 * r};
}

__global__ __launch_bounds__(256) void ce_dice_fwd_kernel(const float* logits, const int64_t* labels, const float* cw,
                                                          float* part, int HW, int ignore_index) {
  const int b = blockIdx.y, sp = blockIdx.x, S = gridDim.x;
  const int per = (HW + S - 1) / S;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2];
  const float one = 1.0f + 1e-6f, eps = 1e-6f;     // fp32 one-hot(+eps) of dice.py:59
  float wn = 0.f, ws = 0.f, it = 0.f, cd = 0.f;
  const float* lg = logits + (int64_t)b * 3 * HW;
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const float a = lg[p], bb = lg[HW + p], c = lg[2 * HW + p];
    const int64_t t = labels[(int64_t)b * HW + p];
    float lse;
    const Px3 q = softmax3(a, bb, c, lse);
    const bool valid = t != ignore_index;
    const int t0 = valid ? (int)t : 0;
    const float xt = t0 == 0 ? a : (t0 == 1 ? bb : c);
    const float pt = t0 == 0 ? q.p0 : (t0 == 1 ? q.p1 : q.p2);
    if (valid) { const float w = t0 == 0 ? w0 : (t0 == 1 ? w1 : w2); wn += w * (lse - xt); ws += w; }
    const float psum = q.p0 + q.p1 + q.p2;
    it += pt * one + (psum - pt) * eps;
    cd += psum + one + 2.f * eps;
  }
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wn = wave_sum(wn); ws = wave_sum(ws); it = wave_sum(it); cd = wave_sum(cd);
  if (lane == 0) { red[wave][0] = wn; red[wave][1] = ws; red[wave][2] = it; red[wave][3] = cd; }
  __syncthreads();
  if (threadIdx.x < 4) part[((size_t)b * S + sp) * 4 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one thread per image walks its S partial rows (fp64, fixed order); thread 0 then adds the per-image terms in image order
// (a single thread walking all B x S rows cost 112 us at B = 32)
__global__ void ce_dice_finish_kernel(const float* part, float* coef, float* out3, int B, int S, int with_dice) {
  __shared__ double swn[64], sws[64], sdice[64];
  const int t = threadIdx.x;
  double wn = 0.0, ws = 0.0, dice = 0.0;
  for (int b = t; b < B; b += 64) {
    double it = 0.0, cd = 0.0;
    for (int sp = 0; sp < S; ++sp) {
      const float* q = part + ((size_t)b * S + sp) * 4;
      wn += q[0]; ws += q[1]; it += q[2]; cd += q[3];
    }
    const double den = cd + 1e-6;
    dice += 1.0 - 2.0 * it / den;
    coef[1 + 2 * b] = (float)(2.0 / den / B);                 // a_b / B
    coef[2 + 2 * b] = (float)(2.0 * it / (den * den) / B);    // b_b / B
  }
  swn[t] = wn; sws[t] = ws; sdice[t] = dice;
  __syncthreads();
  if (t != 0) return;
  wn = 0.0; ws = 0.0; dice = 0.0;
  for (int i = 0; i < 64; ++i) { wn += swn[i]; ws += sws[i]; dice += sdice[i]; }
  const double ce = wn / ws;
  coef[0] = (float)(1.0 / ws);
  dice = with_dice ? dice / B : 0.0;
  out3[0] = (float)(ce + dice); out3[1] = (float)ce; out3[2] = (float)dice;
}

__global__ __launch_bounds__(256) void ce_dice_bwd_kernel(const float* logits, const int64_t* labels, const float* cw,
                                                          const float* coef, const float* gscale, float* dlogits, int HW,
                                                          int ignore_index, int with_dice) {
  const int b = blockIdx.y;
  const float w0 = cw[0], w1 = cw[1], w2 = cw[2];
  const float invw = coef[0], ab = coef[1 + 2 * b], bq = coef[2 + 2 * b];
  const float gs = gscale ? *gscale : 1.f;
  const float one = 1.0f + 1e-6f, eps = 1e-6f;
  const float* lg = logits + (int64_t)b * 3 * HW;
  float* dg = dlogits + (int64_t)b * 3 * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    const float a = lg[p], bb = lg[HW + p], c = lg[2 * HW + p];
    const int64_t t = labels[(int64_t)b * HW + p];
    float lse;
    const Px3 q = softmax3(a, bb, c, lse);
    const bool valid = t != ignore_index;
    const int t0 = valid ? (int)t : 0;
    const float w = valid ? (t0 == 0 ? w0 : (t0 == 1 ? w1 : w2)) * invw : 0.f;
    float g0 = w * (q.p0 - (t0 == 0 ? 1.f : 0.f));
    float g1 = w * (q.p1 - (t0 == 1 ? 1.f : 0.f));
    float g2 = w * (q.p2 - (t0 == 2 ? 1.f : 0.f));
    if (with_dice) {
      const float h0 = -(ab * (t0 == 0 ? one : eps) - bq);
      const float h1 = -(ab * (t0 == 1 ? one : eps) - bq);
      const float h2 = -(ab * (t0 == 2 ? one : eps) - bq);
      const float dot = h0 * q.p0 + h1 * q.p1 + h2 * q.p2;
      g0 += q.p0 * (h0 - dot); g1 += q.p1 * (h1 - dot); g2 += q.p2 * (h2 - dot);
    }
    dg[p] = g0 * gs; dg[HW + p] = g1 * gs; dg[2 * HW + p] = g2 * gs;
  }
}

// argmax (lowest index on ties) + 4x4 confusion matrix
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* logits, const int64_t* labels, int64_t* pred,
                                                               unsigned long long* cm, int C, int HW, int ignore_index) {
  __shared__ unsigned int hist[16];
  if (threadIdx.x < 16) hist[threadIdx.x] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * C * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float m = lg[p]; int am = 0;
    for (int k = 1; k < C; ++k) { const float v = lg[(int64_t)k * HW + p]; if (v > m) { m = v; am = k; } }
    if (pred) pred[(int64_t)b * HW + p] = am;
    const int64_t t = labels[(int64_t)b * HW + p];
    if (t != ignore_index && t >= 0 && t < 4 && am < 4) atomicAdd(&hist[t * 4 + am], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 16 && hist[threadIdx.x]) atomicAdd(&cm[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// the same pass with per-sample group tables (per-AOI / per-climate-zone metrics of the eval loops, change_detection_trainer.py:331-337,
// 437-472): sample b also adds its counts to cms_a[slot_a[b]] and cms_b[slot_b[b]] (slot < 0 or table null: no such group) -- one launch per
// batch instead of one per sample and group
__global__ __launch_bounds__(256) void argmax_confusion_grouped_kernel(const float* logits, const int64_t* labels, int64_t* pred,
                                                                       unsigned long long* cm, const int32_t* slot_a, unsigned long long* cms_a,
                                                                       const int32_t* slot_b, unsigned long long* cms_b, int C, int HW,
                                                                       int ignore_index) {
  __shared__ unsigned int hist[16];
  if (threadIdx.x < 16) hist[threadIdx.x] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const float* lg = logits + (int64_t)b * C * HW;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
    float m = lg[p]; int am = 0;
    for (int k = 1; k < C; ++k) { const float v = lg[(int64_t)k * HW + p]; if (v > m) { m = v; am = k; } }
    if (pred) pred[(int64_t)b * HW + p] = am;
    const int64_t t = labels[(int64_t)b * HW + p];
    if (t != ignore_index && t >= 0 && t < 4 && am < 4) atomicAdd(&hist[t * 4 + am], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 16 && hist[threadIdx.x]) {
    const unsigned long long v = hist[threadIdx.x];
    if (cm) atomicAdd(&cm[threadIdx.x], v);
    if (cms_a && slot_a[b] >= 0) atomicAdd(&cms_a[(int64_t)slot_a[b] * 16 + threadIdx.x], v);
    if (cms_b && slot_b[b] >= 0) atomicAdd(&cms_b[(int64_t)slot_b[b] * 16 + threadIdx.x], v);
  }
}

bool ecam_c_ok(int C, int dtype) {
  const int vec = dtype == KSMI_BF16 ? 8 : 4;
  return C % 16 == 0 && (4 * C) / vec <= kThreads && kThreads % (C / vec) == 0;
}

}  // namespace

#define KSMI_DT(dtype, EXPR_BF16, EXPR_F32)                                  \
  do {                                                                       \
    if ((dtype) == KSMI_BF16) { EXPR_BF16; }                                 \
    else if ((dtype) == KSMI_F32) { EXPR_F32; }                              \
    else return ksmi_fail(KSMI_E_ARG, "bad dtype");                          \
  } while (0)

extern "C" {

static int pool_split() {
  static const int s = ksmi_knob_int("KSMI_ECAM_POOL_SPLIT", 16);
  return s < 1 ? 1 : (s > kPoolSplit ? kPoolSplit : s);
}

size_t ksmi_ecam_pool_workspace(int B, int HW, int C) { (void)HW; return (size_t)B * kPoolSplit * 5 * C * 3 * sizeof(float); }

int ksmi_ecam_pool(const void* const x[4], float* avg, float* mx, int32_t* argmax, float* workspace, int B, int HW, int C,
                   int dtype, void* stream) {
  if (!ecam_c_ok(C, dtype)) return ksmi_fail(KSMI_E_ARG, "ecam_pool: unsupported C");
  const int S = pool_split();
  const size_t n = (size_t)B * S * 5 * C;
  float* psum = workspace; float* pmax = workspace + n; int* pidx = (int*)(workspace + 2 * n);
  KSMI_DT(dtype,
          hipLaunchKernelGGL(ecam_pool_kernel<bf16_t>, dim3(S, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x[0],
                             (const bf16_t*)x[1], (const bf16_t*)x[2], (const bf16_t*)x[3], psum, pmax, pidx, HW, C),
          hipLaunchKernelGGL(ecam_pool_kernel<float>, dim3(S, B), dim3(256), 0, (hipStream_t)stream, (const float*)x[0],
                             (const float*)x[1], (const float*)x[2], (const float*)x[3], psum, pmax, pidx, HW, C));
  int rc = ksmi_check_launch("ecam_pool");
  if (rc) return rc;
  const int tot = B * 5 * C;
  hipLaunchKernelGGL(ecam_pool_final_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, psum, pmax, pidx, avg, mx,
                     argmax, B, S, 5 * C, HW);
  return ksmi_check_launch("ecam_pool_final");
}

int ksmi_ecam_mlp(const float* avg, const float* mx, const float* ca_fc1, const float* ca_fc2, const float* ca1_fc1,
                  const float* ca1_fc2, float* ca, float* ca1, float* hidden, int B, int C, void* stream) {
  if (C % 16) return ksmi_fail(KSMI_E_ARG, "ecam_mlp: C must be a multiple of 16");
  const int HT = (4 * C) / 16 + C / 4;
  hipLaunchKernelGGL(ecam_mlp_kernel, dim3(B), dim3(256), 2 * HT * sizeof(float), (hipStream_t)stream, avg, mx, ca_fc1, ca_fc2,
                     ca1_fc1, ca1_fc2, ca, ca1, hidden, C);
  return ksmi_check_launch("ecam_mlp");
}

int ksmi_ecam_final_forward(const void* const x[4], const float* ca, const float* ca1, const float* wf, const float* bias,
                            float* logits, int B, int HW, int C, int ncls, int dtype, void* stream) {
  if (ncls != 3 || !ecam_c_ok(C, dtype)) return ksmi_fail(KSMI_E_UNSUPPORTED, "ecam_final: ncls must be 3, C multiple of 16");
  const size_t lds = (size_t)(3 * 4 * C + 3) * sizeof(float);
  const dim3 grid((HW + 255) / 256 > 256 ? 256 : (HW + 255) / 256, B);
  KSMI_DT(dtype,
          hipLaunchKernelGGL((ecam_final_fwd_kernel<bf16_t, 3>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)x[0],
                             (const bf16_t*)x[1], (const bf16_t*)x[2], (const bf16_t*)x[3], ca, ca1, wf, bias, logits, HW, C),
          hipLaunchKernelGGL((ecam_final_fwd_kernel<float, 3>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)x[0],
                             (const float*)x[1], (const float*)x[2], (const float*)x[3], ca, ca1, wf, bias, logits, HW, C));
  return ksmi_check_launch("ecam_final_fwd");
}

size_t ksmi_ecam_bwd_workspace(int B, int HW, int C, int ncls) {
  (void)HW;
  const size_t pg = (size_t)B * kBwdSplit * ncls * 4 * C, pd = (size_t)B * kBwdSplit * ncls;
  const size_t gb = (size_t)B * ncls * 4 * C, db = (size_t)B * ncls;
  const size_t mlp = (size_t)B * 5 * C + (size_t)B * 2 * ((4 * C) / 16 + C / 4);
  return (pg + pd + gb + db + mlp) * sizeof(float);
}

int ksmi_ecam_final_backward_reduce(const void* const x[4], const float* dlogits, const float* ca, const float* ca1,
                                    const float* wf, float* dca, float* dca1, float* dwf, float* dbias, float* workspace,
                                    int B, int HW, int C, int ncls, int dtype, void* stream) {
  if (ncls != 3 || !ecam_c_ok(C, dtype)) return ksmi_fail(KSMI_E_UNSUPPORTED, "ecam_final_bwd: ncls must be 3");
  float* pg = workspace; float* pd = pg + (size_t)B * kBwdSplit * 3 * 4 * C;
  float* gb = pd + (size_t)B * kBwdSplit * 3; float* db = gb + (size_t)B * 3 * 4 * C;
  KSMI_DT(dtype,
          hipLaunchKernelGGL((ecam_bwd_reduce_kernel<bf16_t, 3>), dim3(kBwdSplit, B), dim3(256), 0, (hipStream_t)stream,
                             (const bf16_t*)x[0], (const bf16_t*)x[1], (const bf16_t*)x[2], (const bf16_t*)x[3], dlogits, pg, pd, HW, C),
          hipLaunchKernelGGL((ecam_bwd_reduce_kernel<float, 3>), dim3(kBwdSplit, B), dim3(256), 0, (hipStream_t)stream,
                             (const float*)x[0], (const float*)x[1], (const float*)x[2], (const float*)x[3], dlogits, pg, pd, HW, C));
  int rc = ksmi_check_launch("ecam_bwd_reduce");
  if (rc) return rc;
  hipLaunchKernelGGL(ecam_bwd_finish_kernel<3>, dim3(B), dim3(256), (size_t)(3 * 4 * C + 3) * sizeof(float), (hipStream_t)stream, pg, pd, ca, ca1,
                     wf, dca, dca1, gb, db, kBwdSplit, C);
  hipLaunchKernelGGL(ecam_bwd_wf_kernel<3>, dim3((3 * 4 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gb, db, ca, ca1, dwf, dbias, B, C);
  return ksmi_check_launch("ecam_bwd_finish");
}

size_t ksmi_ecam_mlp_bwd_workspace(int B, int C) {
  return ((size_t)B * 5 * C + (size_t)B * 2 * ((4 * C) / 16 + C / 4)) * sizeof(float);
}

int ksmi_ecam_mlp_backward(const float* avg, const float* mx, const float* hidden, const float* ca, const float* ca1,
                           const float* dca, const float* dca1, const float* ca_fc1, const float* ca_fc2, const float* ca1_fc1,
                           const float* ca1_fc2, float* davg, float* dmax, float* g_ca_fc1, float* g_ca_fc2, float* g_ca1_fc1,
                           float* g_ca1_fc2, float* workspace, int B, int C, void* stream) {
  if (C % 16 || !workspace) return ksmi_fail(KSMI_E_ARG, "ecam_mlp_backward: bad args");
  const int C4 = 4 * C, HT = C4 / 16 + C / 4;
  hipLaunchKernelGGL(ecam_mlp_bwd_kernel, dim3(B), dim3(256), (size_t)(5 * C + 2 * HT) * sizeof(float), (hipStream_t)stream, hidden, ca, ca1, dca,
                     dca1, ca_fc1, ca_fc2, ca1_fc1, ca1_fc2, davg, dmax, workspace, B, C);
  const int items = 2 * (C4 * (C4 / 16) + C * (C / 4));
  hipLaunchKernelGGL(ecam_mlp_wgrad_kernel, dim3((items + 255) / 256), dim3(256), 0, (hipStream_t)stream, avg, mx, hidden, workspace, g_ca_fc1,
                     g_ca_fc2, g_ca1_fc1, g_ca1_fc2, B, C);
  return ksmi_check_launch("ecam_mlp_bwd");
}

int ksmi_ecam_final_backward_dx(void* const dx[4], const float* dlogits, const float* ca, const float* wf, const float* davg,
                                const float* dmax, const int32_t* argmax, int B, int HW, int C, int ncls, int dtype, void* stream) {
  if (ncls != 3 || !ecam_c_ok(C, dtype)) return ksmi_fail(KSMI_E_UNSUPPORTED, "ecam_final_bwd_dx: ncls must be 3");
  const size_t lds = (size_t)(3 * 4 * C + 4 * C) * sizeof(float);
  const int tpp = 4 * C / (dtype == KSMI_BF16 ? 8 : 4);          // threads per pixel (source x channel vector)
  if (tpp > 256) return ksmi_fail(KSMI_E_UNSUPPORTED, "ecam_final_bwd_dx: C too large for the (pixel, source, vector) thread map");
  const int ppb = 256 / tpp, trips = 8;                          // ~8 pixel trips per workgroup amortise the coefficient loads
  int gx = (HW + ppb * trips - 1) / (ppb * trips);
  if (gx < 1) gx = 1;
  const dim3 grid(gx, B);
  KSMI_DT(dtype,
          hipLaunchKernelGGL((ecam_bwd_dx_kernel<bf16_t, 3>), grid, dim3(256), lds, (hipStream_t)stream, (bf16_t*)dx[0], (bf16_t*)dx[1],
                             (bf16_t*)dx[2], (bf16_t*)dx[3], dlogits, ca, wf, davg, HW, C),
          hipLaunchKernelGGL((ecam_bwd_dx_kernel<float, 3>), grid, dim3(256), lds, (hipStream_t)stream, (float*)dx[0], (float*)dx[1],
                             (float*)dx[2], (float*)dx[3], dlogits, ca, wf, davg, HW, C));
  int rc = ksmi_check_launch("ecam_bwd_dx");
  if (rc) return rc;
  const int tot = B * C;
  KSMI_DT(dtype,
          hipLaunchKernelGGL(ecam_bwd_max_scatter_kernel<bf16_t>, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                             (bf16_t*)dx[0], (bf16_t*)dx[1], (bf16_t*)dx[2], (bf16_t*)dx[3], dmax, argmax, B, HW, C),
          hipLaunchKernelGGL(ecam_bwd_max_scatter_kernel<float>, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                             (float*)dx[0], (float*)dx[1], (float*)dx[2], (float*)dx[3], dmax, argmax, B, HW, C));
  return ksmi_check_launch("ecam_bwd_max_scatter");
}

size_t ksmi_loss_workspace(int B, int HW) { (void)HW; return ((size_t)B * kLossSplit * 4 + 1 + 2 * (size_t)B) * sizeof(float); }

int ksmi_ce_dice_forward(const float* logits, const int64_t* labels, const float* class_w, int with_dice, float* out3,
                         float* workspace, int B, int HW, int ignore_index, void* stream) {
  if (!logits || !labels || !class_w || !out3 || !workspace || B < 1 || HW < 1) return ksmi_fail(KSMI_E_ARG, "ce_dice_forward: bad args");
  float* part = workspace; float* coef = workspace + (size_t)B * kLossSplit * 4;
  hipLaunchKernelGGL(ce_dice_fwd_kernel, dim3(kLossSplit, B), dim3(256), 0, (hipStream_t)stream, logits, labels, class_w, part, HW, ignore_index);
  hipLaunchKernelGGL(ce_dice_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, coef, out3, B, kLossSplit, with_dice);
  return ksmi_check_launch("ce_dice_fwd");
}

int ksmi_ce_dice_backward(const float* logits, const int64_t* labels, const float* class_w, int with_dice, const float* workspace,
                          const float* grad_scale, float* dlogits, int B, int HW, int ignore_index, void* stream) {
  if (!logits || !labels || !class_w || !dlogits || !workspace) return ksmi_fail(KSMI_E_ARG, "ce_dice_backward: bad args");
  const float* coef = workspace + (size_t)B * kLossSplit * 4;
  const dim3 grid((HW + 255) / 256 > 128 ? 128 : (HW + 255) / 256, B);
  hipLaunchKernelGGL(ce_dice_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, labels, class_w, coef, grad_scale, dlogits, HW,
                     ignore_index, with_dice);
  return ksmi_check_launch("ce_dice_bwd");
}

int ksmi_argmax_confusion(const float* logits, const int64_t* labels, int64_t* pred, int64_t* cm, int B, int C, int HW,
                          int ignore_index, void* stream) {
  if (!logits || !labels || !cm || C < 1 || C > 4) return ksmi_fail(KSMI_E_ARG, "argmax_confusion: 1 <= C <= 4");
  const dim3 grid((HW + 255) / 256 > 128 ? 128 : (HW + 255) / 256, B);
  hipLaunchKernelGGL(argmax_confusion_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, labels, pred, (unsigned long long*)cm, C, HW, ignore_index);
  return ksmi_check_launch("argmax_confusion");
}

int ksmi_argmax_confusion_grouped(const float* logits, const int64_t* labels, int64_t* pred, int64_t* cm, const int32_t* slot_a,
                                  int64_t* cms_a, const int32_t* slot_b, int64_t* cms_b, int B, int C, int HW, int ignore_index,
                                  void* stream) {
  if (!logits || !labels || C < 1 || C > 4) return ksmi_fail(KSMI_E_ARG, "argmax_confusion_grouped: 1 <= C <= 4");
  if ((cms_a && !slot_a) || (cms_b && !slot_b)) return ksmi_fail(KSMI_E_ARG, "argmax_confusion_grouped: a group table without its slot array");
  const dim3 grid((HW + 255) / 256 > 128 ? 128 : (HW + 255) / 256, B);
  hipLaunchKernelGGL(argmax_confusion_grouped_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, labels, pred, (unsigned long long*)cm,
                     slot_a, (unsigned long long*)cms_a, slot_b, (unsigned long long*)cms_b, C, HW, ignore_index);
  return ksmi_check_launch("argmax_confusion_grouped");
}

}  // extern "C"
