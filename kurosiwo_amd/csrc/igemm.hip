// Implicit-GEMM convolution family for gfx950 (MFMA 16x16, LDS-staged halo tiles,
// virtual concat).  See include/ksmi.h for the C-ABI and DESIGN.md for the tiling.
//
//   M = output pixels of one TH x TW patch (<= 256, 4 waves x 4 M-fragments x 16)
//   N = output channels (BN = 16*NT per workgroup)
//   K = taps x channels; channels walk the virtual-concat sources in 64-byte chunks
//       (32 bf16 / 16 fp32), one halo tile + one weight slab in LDS per chunk.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"
#include "igemm_epilogue.h"
#include "wgrad3.h"
#include "igemm3.h"
#include "igemm4.h"

namespace {


template <typename T, int NT, int KH, int KW>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const ksmi_conv_desc d) {
  constexpr int TAPS = KH * KW;
  constexpr int BN = NT * 16;
  constexpr int VEC = ElemTraits<T>::kVec;   // elements per 16 B
  constexpr int KC = VEC * 4;                // elements per 64-byte chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;

  const int tilesX = (d.Wout + d.TW - 1) / d.TW, tilesY = (d.Hout + d.TH - 1) / d.TH;
  int bm = blockIdx.x;
  const int tx = bm % tilesX; bm /= tilesX;
  const int ty = bm % tilesY; const int b = bm / tilesY;
  const int oy0 = ty * d.TH, ox0 = tx * d.TW;
  const int n0 = blockIdx.y * BN;
  const int S = d.stride;
  const int HH = (d.TH - 1) * S + KH, HW = (d.TW - 1) * S + KW;
  const int HP = HH * HW;
  const int P = d.TH * d.TW;
  unsigned char* lds_halo = smem;
  unsigned char* lds_w = smem + ((HP * 64 + 255) & ~255);

  // ---- per-thread halo slots (independent of the chunk) ---------------------------
  constexpr int MAXSLOT = 8;                   // HP*4 <= 2048 vectors
  int slot_goff[MAXSLOT];                      // pixel index in the image, or -1 (zero fill)
  int slot_lds[MAXSLOT];
  const int nvec = HP * 4;
#pragma unroll
  for (int s = 0; s < MAXSLOT; ++s) {
    const int v = tid + s * 256;
    slot_goff[s] = -2; slot_lds[s] = 0;
    if (v < nvec) {
      const int pix = v >> 2, q = v & 3;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int iy = oy0 * S - d.pad + hy, ix = ox0 * S - d.pad_x + hx;
      slot_goff[s] = (iy >= 0 && iy < d.Hin && ix >= 0 && ix < d.Win) ? src_pixel(d, b, iy, ix) : -1;
      slot_lds[s] = pix * 64 + ((q ^ swz(pix)) << 4);
    }
  }
  const int myq = tid & 3;                     // 256 % 4 == 0: a thread always owns k-group myq

  // ---- per-lane A addresses (pixel -> halo position), per tap -------------------------
  int a_addr[4][TAPS];
  bool pvalid[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    int p = wave * 64 + mf * 16 + l15;
    pvalid[mf] = p < P;
    if (p >= P) p = 0;
    const int ly = p / d.TW, lx = p - ly * d.TW;
    const int base = ly * S * HW + lx * S;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int ap = base + (t / KW) * HW + (t % KW);
      a_addr[mf][t] = ap * 64 + ((g ^ swz(ap)) << 4);
    }
  }
  int b_addr[NT];
#pragma unroll
  for (int nf = 0; nf < NT; ++nf) {
    const int n = nf * 16 + l15;
    b_addr[nf] = n * 64 + ((g ^ swz(n)) << 4);
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < NT; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const T* wpk = (const T*)d.wpk;

  for (int ch = 0; ch < d.nchunks; ++ch) {
    const ksmi_src& sr = d.src[chunk_src_of(d, ch)];
    const int c0 = chunk_c0_of(d, ch);
    __syncthreads();
    // ---- stage halo: global -> regs -> (affine, relu) -> LDS ---------------------------
    {
      const int cq = c0 + myq * VEC;           // first channel of my vector (within used range)
      const bool cvalid = cq < sr.c_len;
      float sc[VEC], sh[VEC];
      const bool aff = sr.scale != nullptr;
      if (aff && cvalid) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { sc[j] = sr.scale[cq + j]; sh[j] = sr.shift[cq + j]; }
      }
      const T* sp = (const T*)sr.ptr + sr.c_off + cq;
#pragma unroll
      for (int s = 0; s < MAXSLOT; ++s) {
        if (slot_goff[s] == -2) break;
        u32x4 v = (u32x4){0u, 0u, 0u, 0u};
        if (slot_goff[s] >= 0 && cvalid) {
          v = *(const u32x4*)(sp + (size_t)slot_goff[s] * sr.C);
          if (aff) {
            float f[VEC];
            vec_unpack<T>(v, f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              f[j] = f[j] * sc[j] + sh[j];
              if (sr.relu) f[j] = fmaxf(f[j], 0.f);
            }
            v = vec_pack<T>(f);
          }
        }
        *(u32x4*)(lds_halo + slot_lds[s]) = v;
      }
    }
    // ---- stage weights [TAPS][BN][KC] ---------------------------------------------------
    {
      const T* wsrc = wpk + (size_t)ch * TAPS * d.Npad * KC;
      for (int v = tid; v < TAPS * BN * 4; v += 256) {
        const int row = v >> 2, q = v & 3;
        const int t = row / BN, n = row - t * BN;
        u32x4 w = (u32x4){0u, 0u, 0u, 0u};
        if (n0 + n < d.Npad) w = *(const u32x4*)(wsrc + ((size_t)t * d.Npad + n0 + n) * KC + q * VEC);
        *(u32x4*)(lds_w + row * 64 + ((q ^ swz(n)) << 4)) = w;
      }
    }
    __syncthreads();
    // ---- MFMA ------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      u32x4 a[4], bb[NT];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) a[mf] = *(const u32x4*)(lds_halo + a_addr[mf][t]);
#pragma unroll
      for (int nf = 0; nf < NT; ++nf) bb[nf] = *(const u32x4*)(lds_w + t * BN * 64 + b_addr[nf]);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NT; ++nf) mma16<T>(acc[mf][nf], a[mf], bb[nf]);
    }
  }

  igemm_epilogue<T, NT>(d, acc, smem, tid, wave, g, l15, b, oy0, ox0, n0, P);
}

// -------------------------------------------------------------------------------------------------
// weight packing
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weights_kernel(const ksmi_pack_desc d) {
  constexpr int KC = ElemTraits<T>::kVec * 4;
  const size_t total = (size_t)d.nchunks * d.taps * d.Npad * KC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kk = i % KC; size_t r = i / KC;
    const int j = r % d.Npad; r /= d.Npad;
    const int tap = r % d.taps; const int ch = r / d.taps;
    float v = 0.f;
    const int koff = d.uniform_kc ? ch * d.uniform_kc : d.k_off[ch];
    const int klen = d.uniform_kc ? min(d.uniform_kc, d.k_total - koff) : d.k_len[ch];
    if (j < d.N && kk < klen) {
      const int64_t k = koff + kk;
      const int tp = d.use_tap_map ? d.tap_map[tap] : (d.flip ? (d.taps - 1 - tap) : tap);
      if (tp >= 0) v = d.w[k * d.sK + (int64_t)(j % d.n_mod) * d.sN + (int64_t)(j / d.n_mod) * d.sD + tp * d.sT];
    }
    ElemTraits<T>::st((T*)d.out + i, v);
  }
}

// all packs of a plan in ONE launch: descs live in device memory, blockIdx.y selects the descriptor
template <typename T>
__global__ void pack_weights_batched_kernel(const ksmi_pack_desc* descs) {
  // one 16-byte vector (8 bf16 / 4 fp32 consecutive k of one packed row) per thread and trip, 32-bit index arithmetic (a packed
  // tensor has < 2^31 elements; the first version walked single elements with 64-bit div / mod: 188 us per SNUNet step)
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int KC = VEC * 4;
  const ksmi_pack_desc& d = descs[blockIdx.y];
  const int nchunks = d.nchunks, taps = d.taps, Npad = d.Npad, N = d.N, n_mod = d.n_mod, flip = d.flip;
  const int64_t sK = d.sK, sN = d.sN, sD = d.sD, sT = d.sT;
  const float* w = d.w; T* out = (T*)d.out;
  const uint32_t nvec = (uint32_t)nchunks * (uint32_t)taps * (uint32_t)Npad * 4u;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
    const uint32_t q = v & 3u; uint32_t r = v >> 2;
    const uint32_t j = r % (uint32_t)Npad; r /= (uint32_t)Npad;
    const uint32_t tap = r % (uint32_t)taps; const uint32_t ch = r / (uint32_t)taps;
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = 0.f;
    const int koff = d.uniform_kc ? (int)ch * d.uniform_kc : d.k_off[ch];
    const int klen = d.uniform_kc ? min(d.uniform_kc, d.k_total - koff) : d.k_len[ch];
    const int tp = d.use_tap_map ? d.tap_map[tap] : (flip ? (taps - 1 - (int)tap) : (int)tap);
    if ((int)j < N && tp >= 0) {
      const float* base = w + (int64_t)(j % (uint32_t)n_mod) * sN + (int64_t)(j / (uint32_t)n_mod) * sD + (int64_t)tp * sT;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int kk = (int)q * VEC + e;
        if (kk < klen) f[e] = base[(int64_t)(koff + kk) * sK];
      }
    }
    *(u32x4*)(out + (size_t)v * VEC) = vec_pack<T>(f);
  }
}

// -------------------------------------------------------------------------------------------------
// weight gradient: per (pixel-split, chunk, N-tile) workgroup accumulates
//   G[tap][kc][n] = sum_p X[p*S + tap - pad][kc] * dY[p][n]  over its patches.
// MFMA rows = channels of the chunk, cols = n, K = pixels; both operands are read
// "transposed" from LDS (ds_read_b64_tr_b16 for bf16, scalar ds_read_b32 for fp32).
// -------------------------------------------------------------------------------------------------
template <typename T> struct TrRead;
template <> struct TrRead<bf16_t> {
  // rows = 8 consecutive k (pixels) of k-group g; returns the lane's 8 bf16 at column col0+l15.
  // rowaddr(j) gives the LDS byte address of pixel-row (g*8+j), already including col0*2.
  template <typename F> __device__ static __forceinline__ u32x4 read(F rowaddr, int l15) {
    const int jr = l15 >> 2, q = l15 & 3;
    const unsigned a0 = rowaddr(jr) + q * 8, a1 = rowaddr(4 + jr) + q * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
    u32x4 r;
    r[0] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    r[1] = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    r[2] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    r[3] = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return r;
  }
};

// LDS images of the weight-gradient kernel, swizzled at 32-byte granules (= one 16-column fragment row) so that
// the 2 x 32 lanes of a ds_read_b64_tr_b16 (4 rows x 32 B per 16-lane group; rows p..p+3 and p+8..p+11) land on
// 8 distinct 32-byte bank groups (unswizzled: 2-way on the X tile, 4-way on the dY tile; rocprofv3 showed
// SQ_LDS_BANK_CONFLICT = 58 % of SQ_LDS_IDX_ACTIVE):
//   X halo tile [halo pixel][64 B]        granule' = granule ^ (((hx >> 3) ^ hy) & 1)   (hy, hx = halo row / column: for patches
//                                          whose width divides 16 the term is the same for every pixel k-step of a lane)
//   dY tile     [pixel row][GR * 32 B]    granule' = granule ^ f(row),  f = (row>>1 & 1) | (row>>3 & 1) << 1   (row & 7 for GR = 8)
__device__ __forceinline__ int wg_x_off(int hp, int hy, int hx, int byte) {
  return hp * 64 + ((((byte >> 5) ^ (hx >> 3) ^ hy) & 1) << 5) + (byte & 31);
}
template <int GR>
__device__ __forceinline__ int wg_dy_off(int row, int byte) {
  const int f = GR >= 8 ? (row & 7) : ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) & (GR - 1));
  return row * (GR * 32) + ((((byte >> 5) ^ f) & (GR - 1)) << 5) + (byte & 31);
}

template <typename T, int NT, int KH, int KW, bool LIN>
__global__ __launch_bounds__(256, ((NT == 4 && KH * KW >= 9) ? 1 : 2)) void igemm_wgrad_kernel(const ksmi_wgrad_desc d, int patches_total, int patches_per_split, int dbg) {
  constexpr int TAPS = KH * KW;
  constexpr int BN = NT * 16;
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int KC = VEC * 4;
  constexpr int CF = KC / 16;                  // channel fragments per chunk (2 bf16, 1 fp32)
  constexpr int KSTEP = KC;                    // pixels per MFMA k-step (32 bf16 / 16 fp32)
  constexpr int TPW = (TAPS + 3) / 4;          // taps per wave
  constexpr bool SPLITK = TAPS == 1;
  constexpr int ES = sizeof(T);
  constexpr int GR = (BN * ES / 32) > 0 ? (BN * ES / 32) : 1;   // 32-byte granules per dY row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int split = blockIdx.x, ch = blockIdx.y, n0 = blockIdx.z * BN;
  const int S = d.stride;
  const int HH = (d.TH - 1) * S + KH, HW = (d.TW - 1) * S + KW;
  const int HP = HH * HW;
  const int P = d.TH * d.TW;
  const int Ppad = (P + KSTEP - 1) / KSTEP * KSTEP;
  const int tilesX = (d.Wout + d.TW - 1) / d.TW, tilesY = (d.Hout + d.TH - 1) / d.TH;
  const FastDiv dTW(d.TW), dHW(HW);
  // LDS: X halo [HP][KC] (64 B rows, unswizzled) ; dY tile [Ppad][BN] (BN*ES bytes per row)
  unsigned char* lds_x = smem;
  unsigned char* lds_dy = smem + ((HP * 64 + 255) & ~255);
  const int dyrow = BN * ES;

  const ksmi_src& sr = d.src[d.uniform_kc ? 0 : d.chunk_src[ch]];
  const int c0 = d.uniform_kc ? ch * d.uniform_kc : d.chunk_c0[ch];
  const int myq = tid & 3;
  const int cq = c0 + myq * VEC;
  const bool cvalid = cq < sr.c_len;
  float sc[VEC], sh[VEC];
  const bool aff = sr.scale != nullptr;
  const bool aff_relu = sr.relu != 0;          // (kept in a register: a reference into the kernarg is re-fetched at every use)
  if (aff && cvalid) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sc[j] = sr.scale[cq + j]; sh[j] = sr.shift[cq + j]; }
  }

  f32x4 acc[TPW][CF][NT];
#pragma unroll
  for (int a = 0; a < TPW; ++a)
#pragma unroll
    for (int c = 0; c < CF; ++c)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[a][c][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- patch-invariant per-thread tables (the patch loop then needs a handful of VALU per 16-byte vector) ------------
  constexpr int MAXS = 8;                      // HP*4 <= 2048 vectors, 256 threads
  constexpr int VPR = BN / VEC;                // 16-byte vectors per dY row = dY vectors per thread (Ppad <= 256)
  const int padx = d.pad_x_set ? d.pad_x : d.pad;
  const uint32_t xcb = (uint32_t)sr.C * ES, dycb = (uint32_t)d.dyC * ES;
  const int x_sy = d.in_sy == 0 ? d.Win : d.in_sy * d.in_W, x_sx = d.in_sy == 0 ? 1 : d.in_sx;
  // slot s of a thread: X halo vector tid + 256 s -> halo pixel (tid >> 2) + 64 s; dY vector of pixel tid / VPR + s (256 / VPR).
  // (hy, hx) / (ly, lx) are recomputed per patch from two multiplies: cheaper than 16 table registers on the NT = 4 tile.
  auto x_slot = [&](int s_, int& hy, int& hx) -> bool {
    const int pix = (tid >> 2) + s_ * 64;
    hy = dHW.div(pix); hx = pix - hy * HW;
    return pix < HP;
  };
  const int dq = tid % VPR;                    // 256 % VPR == 0: a thread always owns vector dq of its dY rows
  const bool dqvalid = n0 + dq * VEC < d.N;
  auto dy_slot = [&](int s_, int& ly, int& lx) -> int {      // 1: pixel row, 0: zero row of the k padding, -1: no vector
    const int pp = tid / VPR + s_ * (256 / VPR);
    ly = dTW.div(pp); lx = pp - ly * d.TW;
    return pp < Ppad ? (pp < P ? 1 : 0) : -1;
  };
  const unsigned char* const xsp = (const unsigned char*)((const T*)sr.ptr + sr.c_off + cq);
  const unsigned char* const dyp = (const unsigned char*)((const T*)d.dy + d.dy_c_off + n0 + dq * VEC);

  // bf16 fragment address tables of k-step 0 (dY: the swizzle term repeats every 32 rows; X: see LIN above)
  unsigned b0_tab[sizeof(T) == 2 ? NT : 1][2];
  unsigned a0_tab[(LIN && sizeof(T) == 2) ? TPW : 1][2];      // channel fragment 0; fragment 1 is the other 32-byte half (^ 32)
  unsigned lin_stride = 0;
  if constexpr (sizeof(T) == 2) {
    const unsigned ldx0 = (unsigned)(uintptr_t)lds_x + (l15 & 3) * 8, ldy0 = (unsigned)(uintptr_t)lds_dy + (l15 & 3) * 8;
#pragma unroll
    for (int nf = 0; nf < NT; ++nf)
#pragma unroll
      for (int h = 0; h < 2; ++h) b0_tab[nf][h] = ldy0 + wg_dy_off<GR>(g * 8 + h * 4 + (l15 >> 2), nf * 32);
    if constexpr (LIN) {
      lin_stride = (unsigned)((KSTEP / d.TW) * S * HW * 64);
#pragma unroll
      for (int a_ = 0; a_ < TPW; ++a_) {
        const int t = SPLITK ? 0 : min(wave + a_ * 4, TAPS - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pp = g * 8 + h * 4 + (l15 >> 2);
          const int ly = dTW.div(pp), lx = pp - ly * d.TW;
          const int hy = ly * S + t / KW, hx = lx * S + t % KW;
          a0_tab[a_][h] = ldx0 + wg_x_off(hy * HW + hx, hy, hx, 0);
        }
      }
    }
  }

  const int p_begin = split * patches_per_split;
  const int p_end = min(patches_total, p_begin + patches_per_split);
  // Software pipeline over patches: the global loads of patch i+1 are issued before the MFMAs of patch i and land in registers
  // while they run (a workgroup walks 6-25 patches and only 1-3 workgroups share a CU: nothing else hides the load latency).
  u32x4 xv[MAXS], dv[VPR];
  auto patch_origin = [&](int patch, int& b, int& oy0, int& ox0) {
    int bm = patch;
    const int tx = bm % tilesX; bm /= tilesX;
    const int ty = bm % tilesY; b = bm / tilesY;
    oy0 = ty * d.TH; ox0 = tx * d.TW;
  };
  auto load_patch = [&](int patch) {
    int b, oy0, ox0;
    patch_origin(patch, b, oy0, ox0);
    const int iy0 = oy0 * S - d.pad, ix0 = ox0 * S - padx;
    const int xbase = d.in_sy == 0 ? (b * d.Hin + iy0) * d.Win + ix0
                                   : (b * d.in_H + iy0 * d.in_sy + d.in_oy) * d.in_W + ix0 * d.in_sx + d.in_ox;
    const int dybase = (b * d.Hout + oy0) * d.Wout + ox0;
#pragma unroll
    for (int s_ = 0; s_ < MAXS; ++s_) {
      xv[s_] = (u32x4){0u, 0u, 0u, 0u};
      if (s_ * 256 < HP * 4) {
        int hy, hx;
        const bool in = x_slot(s_, hy, hx);
        const int iy = iy0 + hy, ix = ix0 + hx;
        if (in && cvalid && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win)
          xv[s_] = *(const u32x4*)(xsp + (uint32_t)(xbase + hy * x_sy + hx * x_sx) * xcb);
      }
    }
#pragma unroll
    for (int s_ = 0; s_ < VPR; ++s_) {
      dv[s_] = (u32x4){0u, 0u, 0u, 0u};
      int ly, lx;
      const int kind = dy_slot(s_, ly, lx);
      if (kind > 0 && dqvalid && oy0 + ly < d.Hout && ox0 + lx < d.Wout)
        dv[s_] = *(const u32x4*)(dyp + (uint32_t)(dybase + ly * d.Wout + lx) * dycb);
    }
  };
  auto store_patch = [&](int patch) {
    int b, oy0, ox0;
    patch_origin(patch, b, oy0, ox0);
    const int iy0 = oy0 * S - d.pad, ix0 = ox0 * S - padx;
#pragma unroll
    for (int s_ = 0; s_ < MAXS; ++s_)
      if (s_ * 256 < HP * 4) {
        int hy, hx;
        if (!x_slot(s_, hy, hx)) continue;
        if (aff) {
          if (cvalid && (unsigned)(iy0 + hy) < (unsigned)d.Hin && (unsigned)(ix0 + hx) < (unsigned)d.Win) {
            float f[VEC];
            vec_unpack<T>(xv[s_], f);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              f[j] = f[j] * sc[j] + sh[j];
              if (aff_relu) f[j] = fmaxf(f[j], 0.f);
            }
            xv[s_] = vec_pack<T>(f);
          }
        }
        *(u32x4*)(lds_x + wg_x_off((tid >> 2) + s_ * 64, hy, hx, myq * 16)) = xv[s_];
      }
#pragma unroll
    for (int s_ = 0; s_ < VPR; ++s_) {
      int ly, lx;
      if (dy_slot(s_, ly, lx) >= 0) *(u32x4*)(lds_dy + wg_dy_off<GR>(tid / VPR + s_ * (256 / VPR), dq * 16)) = dv[s_];
    }
  };
  if (p_begin < p_end) load_patch(p_begin);
  for (int patch = p_begin; patch < p_end; ++patch) {
    __syncthreads();                           // the previous patch's MFMAs are done with the LDS images
    if (!(dbg & 4)) store_patch(patch);
    __syncthreads();
    if (patch + 1 < p_end && !(dbg & 2)) load_patch(patch + 1);
    if (dbg & 1) continue;
    // ---- MFMA over pixel k-steps --------------------------------------------------------------
    // 1x1 (token GEMM) case: a single tap would keep one wave busy, so the four waves split the pixel k-steps instead
    // and their accumulators are summed through LDS after the patch loop (SPLITK)
    if constexpr (sizeof(T) == 2) {
      const unsigned ldx = (unsigned)(uintptr_t)lds_x + (l15 & 3) * 8, ldy = (unsigned)(uintptr_t)lds_dy + (l15 & 3) * 8;
      for (int ks = 0; ks < Ppad; ks += KSTEP) {
        if (SPLITK && (((ks / KSTEP) & 3) != wave)) continue;
        // halo pixel of the two transposed-read rows (jr, 4 + jr) of this k-step.  LIN (patch width divides 16, whole k-steps):
        // a lane's column and row parity are the same in every k-step, so its addresses are the k-step-0 table plus a scalar.
        int xa[2], xhy[2], xhx[2];
        const unsigned a_ks = LIN ? (unsigned)(ks / KSTEP) * lin_stride : 0u;
        if constexpr (!LIN) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int pp = ks + g * 8 + h * 4 + (l15 >> 2);
            if (pp >= P) pp = 0;               // dY rows >= P are zero
            const int ly = dTW.div(pp), lx = pp - ly * d.TW;
            xa[h] = ly * S * HW + lx * S; xhy[h] = ly * S; xhx[h] = lx * S;
          }
        }
        u32x4 bfrag[NT];
#pragma unroll
        for (int nf = 0; nf < NT; ++nf) {
          const unsigned a0 = b0_tab[nf][0] + (unsigned)ks * (GR * 32), a1 = b0_tab[nf][1] + (unsigned)ks * (GR * 32);
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
          bfrag[nf] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        // NT <= 2: every fragment of the k-step is requested before the first MFMA (one LDS round trip per k-step instead of
        // one per (tap, channel fragment) group); the NT = 4 tile has no registers left for that
        constexpr bool BATCH = NT <= 2;
        u32x4 afrag[BATCH ? TPW : 1][BATCH ? CF : 1];
        auto read_a = [&](int a_, int cf) -> u32x4 {
          unsigned a0, a1;
          if constexpr (LIN) {
            a0 = (a0_tab[a_][0] ^ (unsigned)(cf * 32)) + a_ks; a1 = (a0_tab[a_][1] ^ (unsigned)(cf * 32)) + a_ks;
          } else {
            const int t = SPLITK ? 0 : wave + a_ * 4;
            const int toff = (t / KW) * HW + (t % KW);
            a0 = ldx + wg_x_off(xa[0] + toff, xhy[0] + t / KW, xhx[0] + t % KW, cf * 32);
            a1 = ldx + wg_x_off(xa[1] + toff, xhy[1] + t / KW, xhx[1] + t % KW, cf * 32);
          }
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
          return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        if constexpr (BATCH) {
#pragma unroll
          for (int a_ = 0; a_ < TPW; ++a_) {
            if ((SPLITK ? 0 : wave + a_ * 4) >= TAPS) break;
#pragma unroll
            for (int cf = 0; cf < CF; ++cf) afrag[a_][cf] = read_a(a_, cf);
          }
        }
#pragma unroll
        for (int a_ = 0; a_ < TPW; ++a_) {
          if ((SPLITK ? 0 : wave + a_ * 4) >= TAPS) break;
#pragma unroll
          for (int cf = 0; cf < CF; ++cf) {
            u32x4 af;
            if constexpr (BATCH) af = afrag[a_][cf]; else af = read_a(a_, cf);
#pragma unroll
            for (int nf = 0; nf < NT; ++nf) mma16<T>(acc[a_][cf][nf], af, bfrag[nf]);
          }
        }
      }
    } else {
      for (int ks = 0; ks < Ppad; ks += KSTEP) {
        if (SPLITK && (((ks / KSTEP) & 3) != wave)) continue;
        u32x4 bfrag[NT];
#pragma unroll
        for (int nf = 0; nf < NT; ++nf)
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
            bfrag[nf][s_] = *(const uint32_t*)(lds_dy + wg_dy_off<GR>(ks + g * 4 + s_, (nf * 16 + l15) * 4));
#pragma unroll
        for (int a_ = 0; a_ < TPW; ++a_) {
          const int t = SPLITK ? 0 : wave + a_ * 4;
          if (t >= TAPS) break;
          const int toff = (t / KW) * HW + (t % KW);
#pragma unroll
          for (int cf = 0; cf < CF; ++cf) {
            u32x4 afrag;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
              int pp = ks + g * 4 + s_; if (pp >= P) pp = 0;
              const int ly = dTW.div(pp), lx = pp - ly * d.TW;
              afrag[s_] = *(const uint32_t*)(lds_x + wg_x_off(ly * S * HW + lx * S + toff, ly * S + t / KW, lx * S + t % KW, l15 * 4));
            }
#pragma unroll
            for (int nf = 0; nf < NT; ++nf) mma16<T>(acc[a_][cf][nf], afrag, bfrag[nf]);
          }
        }
      }
    }
  }
  if (SPLITK) {
    __syncthreads();
    float* red = (float*)smem;                       // [4 waves][CF][NT][4 regs][64 lanes]
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
      for (int nf = 0; nf < NT; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((wave * CF + cf) * NT + nf) * 4 + r) * 64 + lane] = acc[0][cf][nf][r];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
      for (int nf = 0; nf < NT; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sacc = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) sacc += red[(((w * CF + cf) * NT + nf) * 4 + r) * 64 + lane];
          acc[0][cf][nf][r] = sacc;
        }
  }
  // ---- write partial slab [split][tap][chunk*KC + kc][Npad] ------------------------------------
  const int Npad = (d.N + 15) & ~15;
  const int Ktot = d.nchunks * KC;
#pragma unroll
  for (int a = 0; a < TPW; ++a) {
    const int t = wave + a * 4;
    if (t >= TAPS) break;
#pragma unroll
    for (int cf = 0; cf < CF; ++cf)
#pragma unroll
      for (int nf = 0; nf < NT; ++nf) {
        const int n = n0 + nf * 16 + l15;
        if (n >= Npad) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kc = cf * 16 + g * 4 + r;
          d.partial[(((size_t)split * TAPS + t) * Ktot + ch * KC + kc) * Npad + n] = acc[a][cf][nf][r];
        }
      }
  }
}

// sum the split slabs (fp32, fixed summation tree => deterministic) and scatter into the fp32 gradient.
// 8 lanes cooperate on one element (slabs sp = lane8, lane8+8, ...; two independent accumulators each, then a
// 3-step shuffle tree): the serial version (one thread walks up to 256 slabs) cost 44 us per launch.
__global__ void wgrad_reduce_kernel(const ksmi_wgrad_desc d, int taps, int KC) {
  const int Npad = (d.N + 15) & ~15;
  const int Ktot = d.nchunks * KC;
  const size_t total = (size_t)taps * Ktot * Npad;
  const int l8 = threadIdx.x & 7;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < total; i += ((size_t)gridDim.x * blockDim.x) >> 3) {
    const int n = i % Npad; size_t r = i / Npad;
    const int krow = r % Ktot; const int t = r / Ktot;
    const int ch = krow / KC, kc = krow - ch * KC;
    float s0 = 0.f, s1 = 0.f;
    int sp = l8;
    for (; sp + 8 < d.nsplit; sp += 16) { s0 += d.partial[(size_t)sp * total + i]; s1 += d.partial[(size_t)(sp + 8) * total + i]; }
    if (sp < d.nsplit) s0 += d.partial[(size_t)sp * total + i];
    float s = s0 + s1;
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const int koff = d.uniform_kc ? ch * d.uniform_kc : d.k_off[ch];
    const int klen = d.uniform_kc ? min(d.uniform_kc, d.k_total - koff) : d.k_len[ch];
    if (l8 == 0 && n < d.N && kc < klen) {
      const int64_t k = koff + kc;
      float* gp = d.grad + k * d.gK + (int64_t)n * d.gN + (d.use_tap_off ? (int64_t)d.tap_off[t] : (int64_t)t * d.gT);
      *gp = d.accumulate ? (*gp + s) : s;
    }
  }
}

template <typename T>
int launch_fwd(const ksmi_conv_desc* d, hipStream_t st) {
  const int taps = d->KH * d->KW;
  const int tilesX = (d->Wout + d->TW - 1) / d->TW, tilesY = (d->Hout + d->TH - 1) / d->TH;
  const int gm = d->B * tilesX * tilesY;
  const int HH = (d->TH - 1) * d->stride + d->KH, HW = (d->TW - 1) * d->stride + d->KW;
  const int HP = HH * HW;
  if (d->TH * d->TW > 256 || HP * 4 > 2048) return ksmi_fail(KSMI_E_ARG, "conv: patch too large (TH*TW<=256, halo<=512 px)");
  // BN = 16*nt columns per workgroup.  BN=128 would need 94 KB of LDS (1 workgroup per CU, nothing to
  // overlap the single-buffered K loop with): cap at 64 until the main loop is software-pipelined.
  int nt = d->Npad >= 64 ? 4 : (d->Npad >= 32 ? 2 : 1);
  const int bn = nt * 16;
  const dim3 grid(gm, (d->Npad + bn - 1) / bn);
  size_t lds = ((HP * 64 + 255) & ~255) + (size_t)taps * bn * 64;
  const size_t tile = (size_t)d->TH * d->TW * (bn * sizeof(T) + 16);
  if (lds < tile) lds = tile;
  if (lds < 256 * 2 * sizeof(float)) lds = 256 * 2 * sizeof(float);
#define KSMI_LAUNCH_FWD(NT_, KH_, KW_)                                                              \
  do {                                                                                              \
    auto kfn = igemm_fwd_kernel<T, NT_, KH_, KW_>; KSMI_NOTE(kfn);                                                  \
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, *d);                                          \
  } while (0)
#define KSMI_DISPATCH_NT(KH_, KW_)                                                                  \
  switch (nt) {                                                                                     \
    case 1: KSMI_LAUNCH_FWD(1, KH_, KW_); break;                                                    \
    case 2: KSMI_LAUNCH_FWD(2, KH_, KW_); break;                                                    \
    case 4: KSMI_LAUNCH_FWD(4, KH_, KW_); break;                                                    \
    default: KSMI_LAUNCH_FWD(8, KH_, KW_); break;                                                   \
  }
  if (d->KH == 3 && d->KW == 3) { KSMI_DISPATCH_NT(3, 3) }
  else if (d->KH == 1 && d->KW == 1) { KSMI_DISPATCH_NT(1, 1) }
  else if (d->KH == 2 && d->KW == 2) { KSMI_DISPATCH_NT(2, 2) }
  else if (d->KH == 4 && d->KW == 4) { KSMI_DISPATCH_NT(4, 4) }
  else return ksmi_fail(KSMI_E_UNSUPPORTED, "conv: kernel size not supported");
#undef KSMI_DISPATCH_NT
#undef KSMI_LAUNCH_FWD
  return ksmi_check_launch("igemm_fwd");
}


// -------------------------------------------------------------------------------------------------
// Token-GEMM weight gradient (1x1, stride 1, one dense bf16 source): dW^T[cin][n] = sum_rows X[row][cin] * dY[row][n].
// The generic kernel above gives every workgroup a 32 x 64 output tile (21 flop per staged byte); here a workgroup owns a
// 128 x 128 tile (2 x 2 waves of 64 x 64 = 4 x 4 MFMA tiles each) and streams its row slab in steps of 32 rows through a
// double-buffered LDS image (64 flop per byte), global -> register prefetch one step ahead.  Same partial-slab layout and
// reduce kernel as the generic path.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn_wgrad_kernel(const ksmi_wgrad_desc d, int rows_total, int rows_per_split, int K) {
  constexpr int TM = 128, TN = 128, RS = 32;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (RS * TM * 2 + RS * TN * 2)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int k0 = blockIdx.x * TM, n0 = blockIdx.y * TN, split = blockIdx.z;
  const ksmi_src& sr = d.src[0];
  const bf16_t* xp = (const bf16_t*)sr.ptr + sr.c_off;
  const bf16_t* yp = (const bf16_t*)d.dy + d.dy_c_off;
  const int r_begin = split * rows_per_split;
  const int r_end = min(rows_total, r_begin + rows_per_split);
  const int nsteps = (max(r_end - r_begin, 0) + RS - 1) / RS;
  // staging slots: 512 16-byte vectors per tile, 2 per thread: v = tid + i*256 -> row v>>4, vector v&15
  u32x4 rx[2], ry[2];
  auto gload = [&](int step) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + i * 256, r = v >> 4, q = v & 15;
      const int row = r_begin + step * RS + r;
      const bool rok = row < r_end;
      rx[i] = (rok && k0 + q * 8 < sr.c_len) ? *(const u32x4*)(xp + (size_t)row * sr.C + k0 + q * 8) : (u32x4){0u, 0u, 0u, 0u};
      ry[i] = (rok && n0 + q * 8 < d.N) ? *(const u32x4*)(yp + (size_t)row * d.dyC + n0 + q * 8) : (u32x4){0u, 0u, 0u, 0u};
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* bx = smem + buf * (RS * TM * 2 + RS * TN * 2);
    unsigned char* by = bx + RS * TM * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = tid + i * 256, r = v >> 4, q = v & 15;
      *(u32x4*)(bx + wg_dy_off<8>(r, q * 16)) = rx[i];
      *(u32x4*)(by + wg_dy_off<8>(r, q * 16)) = ry[i];
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nsteps > 0) { gload(0); lstore(0); }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) gload(s + 1);
    const unsigned bx = (unsigned)(uintptr_t)(smem + (s & 1) * (RS * TM * 2 + RS * TN * 2));
    const unsigned by = bx + RS * TM * 2;
    u32x4 af[4], bf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      auto ra = [&](int j) -> unsigned { return bx + wg_dy_off<8>(g * 8 + j, (wm * 64 + t * 16) * 2); };
      auto rb = [&](int j) -> unsigned { return by + wg_dy_off<8>(g * 8 + j, (wn * 64 + t * 16) * 2); };
      af[t] = TrRead<bf16_t>::read(ra, l15);
      bf[t] = TrRead<bf16_t>::read(rb, l15);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) mma16<bf16_t>(acc[a][b], af[a], bf[b]);
    if (more) lstore((s + 1) & 1);
    __syncthreads();
  }
  // partial[split][0][krow][Npad]: krow = cin, same as the generic kernel (row = (lane>>4)*4 + r, col = lane&15)
  const int Npad = (d.N + 15) & ~15;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = n0 + wn * 64 + b * 16 + l15;
      if (n >= Npad) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krow = k0 + wm * 64 + a * 16 + g * 4 + r;
        if (krow < K) d.partial[((size_t)split * K + krow) * Npad + n] = acc[a][b][r];
      }
    }
}

// split reduction of the token-GEMM path: float4 per thread along n (coalesced slab reads), fixed summation order
__global__ void tn_reduce_kernel(const ksmi_wgrad_desc d, int KC, int K) {
  const int Npad = (d.N + 15) & ~15, NV = Npad / 4;
  const size_t total = (size_t)K * NV, slab = (size_t)K * Npad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int krow = i / NV, n4 = (int)(i - (size_t)krow * NV) * 4;
    const int ch = krow / KC, kc = krow - ch * KC;
    const int koff = d.uniform_kc ? ch * d.uniform_kc : d.k_off[ch];
    const int klen = d.uniform_kc ? min(d.uniform_kc, d.k_total - koff) : d.k_len[ch];
    if (kc >= klen) continue;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p = d.partial + (size_t)krow * Npad + n4;
    for (int sp = 0; sp < d.nsplit; ++sp) {
      const f32x4 v = *(const f32x4*)(p + (size_t)sp * slab);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int64_t k = koff + kc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n4 + j;
      if (n < d.N) {
        float* g = d.grad + k * d.gK + (int64_t)n * d.gN;
        *g = d.accumulate ? *g + s[j] : s[j];
      }
    }
  }
}

// ... for a PLAIN row-major nn.Linear gradient grad[n][k] (gK == 1): the slabs have n contiguous, the gradient k contiguous, so the
// element-wise reducer above writes every 4-byte value of the gradient into a cache line of its own (4 MB of gradient = one million
// scattered stores: 15.8 us for the 1024 x 1024 layers of the ViT, 0.76 TB/s).  Here a workgroup sums a 32 (k) x 32 (n) tile of the
// slabs with coalesced reads (same split order: bit-identical sums), turns it through LDS and writes 128-byte runs of the gradient.
__global__ __launch_bounds__(256) void tn_reduce_tr_kernel(const float* __restrict__ partial, int nsplit, size_t slab, int Npad, int N, int Kreal,
                                                           float* __restrict__ grad, int64_t gN, int accumulate) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    float sum = 0.f;
    if (k < Kreal && n < Npad) {
      const float* p = partial + (size_t)k * Npad + n;
      for (int sp = 0; sp < nsplit; ++sp) sum += p[(size_t)sp * slab];
    }
    tile[r][tx] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    if (n < N && k < Kreal) {
      float* g = grad + (int64_t)n * gN + k;
      *g = accumulate ? *g + tile[tx][r] : tile[tx][r];
    }
  }
}

static bool wgrad_generic_forced() { static const bool on = ksmi_knob_is_set("KSMI_WGRAD_GENERIC"); return on; }
// eligibility of the token-GEMM path and its split geometry
static bool gemm_tn_eligible(const ksmi_wgrad_desc* d, int es) {
  return es == 2 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->nsrc == 1 && d->src[0].scale == nullptr &&
         d->Hin == d->Hout && d->Win == d->Wout && (d->src[0].c_len % 8) == 0 && (d->N % 8) == 0 && d->N >= 64 && d->src[0].c_len >= 64 &&
         !wgrad_generic_forced();
}
// tile and split choice of the token-GEMM weight gradient (gemm2.hip: 128 A-side columns x `bt` B-side columns per workgroup).
// direct (one split, row-major gradient): A = k, B = n; slabs: A = n, B = k.  Cost = whole rounds of the 256 CUs x 64-row steps
// (measured: 0.95 us per step for bt = 128, ~0.6 for 64) + per-tile prologue/epilogue, + the slab traffic a split costs
// (written once, read once by the reducer at ~4 TB/s) and the reducer launch.
static void gemm_tn_geom(const ksmi_wgrad_desc* d, int kc, int& nsplit, int& rps, int& tk, int& tn, int& bt) {
  const int rows = d->B * d->Hout * d->Wout;
  const int K = d->nchunks * kc;
  const int npad = (d->N + 15) & ~15;
  tk = (K + 127) / 128; tn = (npad + 127) / 128;
  bool plain = d->gK == 1 && (!d->use_tap_off || d->tap_off[0] == 0) && d->gN >= d->src[0].c_len && d->gN < ((int64_t)1 << 31);
  if (plain && !d->uniform_kc)
    for (int i = 0; i < d->nchunks; ++i) plain = plain && d->k_off[i] == i * kc;
  const int steps_all = (rows + 63) / 64;
  double best = 1e30; int bs = 1; bt = 128;
  static const int f_bt = ksmi_knob_int("KSMI_TN_BT", 0), f_s = ksmi_knob_int("KSMI_TN_SPLIT", 0);   // probes
  for (int b = 128; b >= 64; b -= 32) {
    if (f_bt && b != f_bt) continue;
    const double step_us = b == 128 ? 0.95 : b == 96 ? 0.78 : 0.6;
    for (int s = 1; s <= 256 && s <= steps_all; s = s < 8 ? s + 1 : s * 2) {
      if (s == 1 && !plain) continue;
      if (f_s && s != f_s) continue;
      // A-side tiles x B-side tiles (direct: A = k; slabs: A = n)
      const int acols = s == 1 ? K : npad, bcols = s == 1 ? npad : K;
      const int tiles = ((acols + 127) / 128) * ((bcols + b - 1) / b);
      const int steps = (steps_all + s - 1) / s;
      const int rounds = (tiles * s + 255) / 256;
      // a workgroup that would be alone on its CU runs as two wave groups on half the steps each (gemm2_tn_kernel SPL = 2): measured
      // with two co-resident half-range workgroups, a step of the pair costs ~0.73 of two serial ones
      const bool spl = ksmi_gemm2_tn_spl(tiles * s, steps);
      double t = rounds * ((spl ? 0.73 : 1.0) * steps * step_us + (spl ? 4.0 : 3.0));
      if (s > 1) t += 2.0 * s * (double)K * npad * 4.0 / 4.0e6 + 3.0 + 0.08 * (s > 16 ? 16 + (s - 16) / 8 : s);   // + the reducer's serial slab walk
      if (t < best) { best = t; bs = s; bt = b; }
    }
  }
  rps = ((steps_all + bs - 1) / bs) * 64;
  nsplit = (rows + rps - 1) / rps;
}

struct WgradGeom { int taps, kc, nt, bn, patches, pps, nsplit, ntiles, npad; size_t lds; bool tn; int rps, tk, tnn, tbt; bool v3; ksmi_wgrad3_geom_t g3; };

template <typename T>
WgradGeom wgrad_geom(const ksmi_wgrad_desc* d) {
  WgradGeom g;
  g.taps = d->KH * d->KW;
  g.kc = ElemTraits<T>::kVec * 4;
  g.npad = (d->N + 15) & ~15;
  g.nt = g.npad >= 64 ? 4 : (g.npad >= 32 ? 2 : 1);
  static const int wnt_cap = ksmi_knob_int("KSMI_WGRAD_NT", 4);
  if (g.nt > wnt_cap) g.nt = wnt_cap;
  // 3x3 / 4x4: the 64-column tile (96+ accumulators beside the prefetch registers of the patch pipeline) spills at 256 VGPRs and is
  // latency-starved at 512; two 32-column workgroups re-read X from L2 instead and run the same speed or better
  static const bool wnt4 = ksmi_knob_is_set("KSMI_WGRAD_NT4");
  if (g.taps >= 9 && g.nt > 2 && !wnt4) g.nt = 2;
  g.bn = g.nt * 16;
  g.ntiles = (g.npad + g.bn - 1) / g.bn;
  const int tilesX = (d->Wout + d->TW - 1) / d->TW, tilesY = (d->Hout + d->TH - 1) / d->TH;
  g.patches = d->B * tilesX * tilesY;
  // aim at ~1024 workgroups in total; at most 512 splits
  static const int wg_target = ksmi_knob_int("KSMI_WGRAD_WGS", 1024);
  static const int split_cap = ksmi_knob_int("KSMI_WGRAD_SPLITS", 512);   // single-chunk gradients (K <= 32) otherwise run one workgroup per CU
  int want = wg_target / (d->nchunks * g.ntiles);
  if (want < 1) want = 1;
  if (want > split_cap) want = split_cap;
  if (want > g.patches) want = g.patches;
  g.pps = (g.patches + want - 1) / want;
  g.nsplit = (g.patches + g.pps - 1) / g.pps;
  const int HH = (d->TH - 1) * d->stride + d->KH, HW = (d->TW - 1) * d->stride + d->KW;
  const int P = d->TH * d->TW, Ppad = (P + g.kc - 1) / g.kc * g.kc;
  g.lds = ((HH * HW * 64 + 255) & ~255) + (size_t)Ppad * g.bn * sizeof(T);
  if (g.taps == 1 && g.lds < (size_t)(g.kc / 16) * g.nt * 4096) g.lds = (size_t)(g.kc / 16) * g.nt * 4096;   // split-k reduction buffer
  g.tn = gemm_tn_eligible(d, (int)sizeof(T));
  if (g.tn) gemm_tn_geom(d, g.kc, g.nsplit, g.rps, g.tk, g.tnn, g.tbt);
  // 3x3 stride-1 bf16 gradients with whole 32-channel chunks: channel-owner kernel (wgrad3.hip), same slab layout and reducer
  g.v3 = !g.tn && ksmi_wgrad3_geom(d, sizeof(T) == 2 ? KSMI_BF16 : KSMI_F32, &g.g3);
  if (g.v3) g.nsplit = g.g3.nsplit;
  return g;
}

// KSMI_SLAB_BIAS=0: the split-mode token weight gradient leaves the bias gradient to a channel_sum pass (the round-4 route; same-box A/B)
static bool wgrad_slab_bias_on() {
  static const bool on = (ksmi_knob_int("KSMI_SLAB_BIAS", 1) != 0);
  return on;
}

template <typename T>
int launch_wgrad(const ksmi_wgrad_desc* d, hipStream_t st) {
  WgradGeom g = wgrad_geom<T>(d);
  if (d->TH * d->TW > 256) return ksmi_fail(KSMI_E_ARG, "wgrad: patch too large");
  if (d->nsplit != g.nsplit) return ksmi_fail(KSMI_E_ARG, "wgrad: nsplit does not match ksmi_conv_wgrad_workspace geometry");
  if (g.v3) {
    int rc3 = ksmi_wgrad3_launch(d, &g.g3, st);
    if (rc3) return rc3;
    return ksmi_wgrad3_reduce(d, &g.g3, st);
  }
  if (g.tn) {
    // plain row-major nn.Linear gradient (unit K stride, one tap at offset 0, contiguous k rows)
    bool plain = d->gK == 1 && (!d->use_tap_off || d->tap_off[0] == 0) && d->gN >= d->src[0].c_len && d->gN < ((int64_t)1 << 31);
    if (plain && !d->uniform_kc)
      for (int i = 0; i < d->nchunks; ++i) plain = plain && d->k_off[i] == i * g.kc;
    {
      const int rows = d->B * d->Hout * d->Wout;
      const bool direct = plain && g.nsplit == 1;
      // bias_grad: the final gradient in direct mode; in slab mode (plain layers only) the partial column sums of dY per split,
      // [nsplit][N] (ksmi_conv_wgrad_fuses_bias == 2: the caller sums the rows)
      const int r2 = ksmi_gemm2_tn((const bf16_t*)d->src[0].ptr + d->src[0].c_off, d->src[0].C, (const bf16_t*)d->dy + d->dy_c_off, d->dyC, d->partial,
                                   g.npad, direct ? d->grad : nullptr, d->gN, rows, d->src[0].c_len, d->N, d->nchunks * g.kc, g.nsplit, g.rps,
                                   g.tbt, d->accumulate, (direct || (plain && wgrad_slab_bias_on())) ? d->bias_grad : nullptr, d->bias_accumulate, st);
      if (r2 < 0) return r2;
      if (r2 == 0) {
        if (direct) return 0;
        if (g.nsplit > 16) {      // many thin slabs (small matrices): the 8-lanes-per-element tree reducer hides the slab walk better
          const size_t totalr = (size_t)d->nchunks * g.kc * g.npad;
          int blocksr = (int)((totalr * 8 + 255) / 256); if (blocksr > 4096) blocksr = 4096;
          hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocksr), dim3(256), 0, st, *d, 1, g.kc);
          return ksmi_check_launch("wgrad_reduce");
        }
        static const int tr_on = ksmi_knob_int("KSMI_TN_REDUCE_TR", 1);      // 0: the element-wise reducer (round 1)
        if (tr_on && plain) {
          const int Kreal = d->uniform_kc ? d->k_total : d->k_off[d->nchunks - 1] + d->k_len[d->nchunks - 1];
          const dim3 gridr((g.npad + 31) / 32, (Kreal + 31) / 32);
          hipLaunchKernelGGL(tn_reduce_tr_kernel, gridr, dim3(256), 0, st, d->partial, g.nsplit, (size_t)d->nchunks * g.kc * g.npad, g.npad, d->N, Kreal,
                             d->grad, d->gN, d->accumulate);
          return ksmi_check_launch("tn_reduce_tr");
        }
        const size_t total0 = (size_t)d->nchunks * g.kc * g.npad / 4;
        int blocks0 = (int)((total0 + 255) / 256); if (blocks0 > 8192) blocks0 = 8192;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks0), dim3(256), 0, st, *d, g.kc, d->nchunks * g.kc);
        return ksmi_check_launch("tn_reduce");
      }
    }
    hipLaunchKernelGGL(gemm_tn_wgrad_kernel, dim3(g.tk, g.tnn, g.nsplit), dim3(256), 0, st, *d, d->B * d->Hout * d->Wout, g.rps, d->nchunks * g.kc);
    int rc0 = ksmi_check_launch("gemm_tn_wgrad");
    if (rc0) return rc0;
    if (g.nsplit > 16) {      // many thin slabs (small matrices): the 8-lanes-per-element tree reducer hides the slab walk better
      const size_t totalr = (size_t)d->nchunks * g.kc * g.npad;
      int blocksr = (int)((totalr * 8 + 255) / 256); if (blocksr > 4096) blocksr = 4096;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocksr), dim3(256), 0, st, *d, 1, g.kc);
      return ksmi_check_launch("wgrad_reduce");
    }
    const size_t total0 = (size_t)d->nchunks * g.kc * g.npad / 4;
    int blocks0 = (int)((total0 + 255) / 256); if (blocks0 > 8192) blocks0 = 8192;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(blocks0), dim3(256), 0, st, *d, g.kc, d->nchunks * g.kc);
    return ksmi_check_launch("tn_reduce");
  }
  {   // per-lane offsets of the staging loads are 32-bit
    const size_t pin = d->in_sy ? (size_t)d->B * d->in_H * d->in_W : (size_t)d->B * d->Hin * d->Win;
    size_t cmax = d->dyC;
    for (int i = 0; i < d->nsrc; ++i) if ((size_t)d->src[i].C > cmax) cmax = d->src[i].C;
    const size_t pmax = pin > (size_t)d->B * d->Hout * d->Wout ? pin : (size_t)d->B * d->Hout * d->Wout;
    if (pmax * cmax * sizeof(T) >= ((size_t)1 << 32)) return ksmi_fail(KSMI_E_UNSUPPORTED, "wgrad: a tensor of 4 GiB or more is not supported");
  }
  static const int wdbg = ksmi_knob_int("KSMI_WDBG", 0);   // profiling switches: 1 no MFMA, 2 no global loads, 4 no LDS stores
  const dim3 grid(g.nsplit, d->nchunks, g.ntiles);
  // LIN: bf16, patch width divides 16 and the patch is a whole number of 32-pixel k-steps (see the kernel)
  static const bool lin_off = ksmi_knob_is_set("KSMI_WGRAD_NOLIN");
  const bool lin = !lin_off && sizeof(T) == 2 && (16 % d->TW) == 0 && ((d->TH * d->TW) % 32) == 0;
#define KSMI_LAUNCH_WG(NT_, KH_, KW_)                                                               \
  do {                                                                                              \
    if (lin) {                                                                                      \
      auto kfn = igemm_wgrad_kernel<T, NT_, KH_, KW_, true>; KSMI_NOTE(kfn);                                        \
      if (g.lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); \
      hipLaunchKernelGGL(kfn, grid, dim3(256), g.lds, st, *d, g.patches, g.pps, wdbg);              \
    } else {                                                                                        \
      auto kfn = igemm_wgrad_kernel<T, NT_, KH_, KW_, false>; KSMI_NOTE(kfn);                                       \
      if (g.lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds); \
      hipLaunchKernelGGL(kfn, grid, dim3(256), g.lds, st, *d, g.patches, g.pps, wdbg);              \
    }                                                                                               \
  } while (0)
#define KSMI_DISPATCH_WNT(KH_, KW_)                                                                 \
  switch (g.nt) {                                                                                   \
    case 1: KSMI_LAUNCH_WG(1, KH_, KW_); break;                                                     \
    case 2: KSMI_LAUNCH_WG(2, KH_, KW_); break;                                                     \
    default: KSMI_LAUNCH_WG(4, KH_, KW_); break;                                                    \
  }
  if (d->KH == 3 && d->KW == 3) { KSMI_DISPATCH_WNT(3, 3) }
  else if (d->KH == 1 && d->KW == 1) { KSMI_DISPATCH_WNT(1, 1) }
  else if (d->KH == 2 && d->KW == 2) { KSMI_DISPATCH_WNT(2, 2) }
  else if (d->KH == 4 && d->KW == 4) { KSMI_DISPATCH_WNT(4, 4) }
  else return ksmi_fail(KSMI_E_UNSUPPORTED, "wgrad: kernel size not supported");
#undef KSMI_DISPATCH_WNT
#undef KSMI_LAUNCH_WG
  int rc = ksmi_check_launch("igemm_wgrad");
  if (rc) return rc;
  const size_t total = (size_t)g.taps * d->nchunks * g.kc * g.npad;
  int blocks = (int)((total * 8 + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, *d, g.taps, g.kc);
  return ksmi_check_launch("wgrad_reduce");
}

}  // namespace

extern "C" {

int ksmi_chunk_elems(int dtype) { return dtype == KSMI_BF16 ? 32 : 16; }

int ksmi_conv_grid_m(const ksmi_conv_desc* d) {
  const int tilesX = (d->Wout + d->TW - 1) / d->TW, tilesY = (d->Hout + d->TH - 1) / d->TH;
  return d->B * tilesX * tilesY;
}

size_t ksmi_desc_size(int which) {
  switch (which) {
    case 0: return sizeof(ksmi_conv_desc);
    case 1: return sizeof(ksmi_wgrad_desc);
    case 2: return sizeof(ksmi_pack_desc);
    case 3: return sizeof(ksmi_rowsum_desc);
    case 4: return sizeof(ksmi_tiff_info);
    default: return 0;
  }
}

int ksmi_conv_gate_supported(const ksmi_conv_desc* d, int dtype) {
  if (!d) return 0;
  ksmi_conv_desc c = *d;
  if (!c.stats) c.stats = (float*)(uintptr_t)16;
  if (!c.gate_src) { c.gate_src = (const void*)(uintptr_t)16; c.xhat_src = c.gate_src; c.g_mean = (const float*)(uintptr_t)16; c.g_rstd = c.g_mean; }
  ksmi_igemm4_geom_t g4;
  return ksmi_igemm4_geom(&c, dtype, &g4) ? 1 : 0;
}

int ksmi_conv_stats_rows(const ksmi_conv_desc* d, int dtype) {
  if (!d) return 0;
  ksmi_conv_desc c = *d;
  if (!c.stats) c.stats = (float*)(uintptr_t)16;     // the answer is for the launch WITH statistics
  ksmi_igemm3_geom_t g3;
  if (!c.gate_src && ksmi_igemm3_geom(&c, dtype, &g3)) return g3.gx;
  ksmi_igemm4_geom_t g4;
  if (ksmi_igemm4_geom(&c, dtype, &g4)) return g4.gx;
  return ksmi_conv_grid_m(d);
}

// Which kernel ksmi_conv_forward(d, dtype) starts, decided on the host (nothing is launched): info[0] = kernel generation (4 igemm4, 3 igemm3,
// 2 igemm2, 1 first generation), info[1] = 1 when that launcher has an instantiated kernel for the geometry it chose (a geometry function
// that accepts a descriptor its launcher cannot start makes the forward raise instead of falling through: ADVICE round 5), igemm4 only:
// info[2..7] = WM, NF, waves per workgroup, schedule (0 row steps, 1 deep ring, 2 chunk), gx, gy.  Test / tooling hook.
int ksmi_conv_dispatch_info(const ksmi_conv_desc* d, int dtype, int32_t* info) {
  if (!d || !info) return ksmi_fail(KSMI_E_ARG, "conv_dispatch_info: null argument");
  for (int i = 0; i < 8; ++i) info[i] = 0;
  static const bool force_v1 = ksmi_knob_is_set("KSMI_IGEMM_V1");
  ksmi_igemm3_geom_t g3;
  if (!force_v1 && !d->gate_src && ksmi_igemm3_geom(d, dtype, &g3) && (d->stats == nullptr || d->stats_rows == g3.gx)) {
    info[0] = 3; info[1] = 1; info[6] = g3.gx; info[7] = g3.gy;
    return 0;
  }
  ksmi_igemm4_geom_t g4;
  if (!force_v1 && ksmi_igemm4_geom(d, dtype, &g4) && (d->stats == nullptr || d->stats_rows == g4.gx)) {
    info[0] = 4; info[1] = ksmi_igemm4_launchable(d, &g4) ? 1 : 0;
    info[2] = g4.WM; info[3] = g4.NF; info[4] = g4.nwv; info[5] = g4.deep; info[6] = g4.gx; info[7] = g4.gy;
    return 0;
  }
  info[0] = (!force_v1 && ksmi_igemm2_eligible(d, dtype)) ? 2 : 1;
  info[1] = 1;
  return 0;
}

int ksmi_conv_forward(const ksmi_conv_desc* d, int dtype, void* stream) {
  if (!d || d->nsrc < 1 || d->nsrc > KSMI_MAX_SRC || d->ndst < 1 || d->ndst > KSMI_MAX_SRC || d->nchunks < 1 ||
      (d->nchunks > KSMI_MAX_CHUNKS && !(d->uniform_kc && d->nsrc == 1)))
    return ksmi_fail(KSMI_E_ARG, "conv: bad descriptor");
  if (dtype != KSMI_BF16 && dtype != KSMI_F32) return ksmi_fail(KSMI_E_ARG, "conv: bad dtype");
  static const bool force_v1 = ksmi_knob_is_set("KSMI_IGEMM_V1");      // A/B switch for profiling
  {   // short K, bf16: persistent workgroups with register-resident weights (its statistics rows = workgroups, not tiles)
    ksmi_igemm3_geom_t g3;
    if (!force_v1 && !d->gate_src && ksmi_igemm3_geom(d, dtype, &g3) && (d->stats == nullptr || d->stats_rows == g3.gx))
      return ksmi_igemm3_launch(d, &g3, (hipStream_t)stream);
    // long K, >= 64 output channels, 3x3: persistent workgroups with halo / weight rings (igemm4.hip); same statistics-row rule
    ksmi_igemm4_geom_t g4;
    if (!force_v1 && ksmi_igemm4_geom(d, dtype, &g4) && (d->stats == nullptr || d->stats_rows == g4.gx))
      return ksmi_igemm4_launch(d, &g4, (hipStream_t)stream);
    if (d->gate_src) return ksmi_fail(KSMI_E_UNSUPPORTED, "conv: the gate epilogue needs the persistent long-K kernel (ksmi_conv_gate_supported)");
  }
  if (d->stats && d->stats_rows != 0 && d->stats_rows != ksmi_conv_grid_m(d))
    return ksmi_fail(KSMI_E_ARG, "conv: stats_rows does not match the kernel this descriptor runs on (set it from ksmi_conv_stats_rows)");
  if (!force_v1 && ksmi_igemm2_eligible(d, dtype)) return ksmi_igemm2_launch(d, dtype, (hipStream_t)stream);
  if (d->alpha != 0.f || d->resid || d->relu_out || d->uniform_kc)
    return ksmi_fail(KSMI_E_UNSUPPORTED, "conv: alpha/resid/relu_out/uniform chunks need whole-chunk sources (igemm2 path)");
  if (dtype == KSMI_BF16) return launch_fwd<bf16_t>(d, (hipStream_t)stream);
  return launch_fwd<float>(d, (hipStream_t)stream);
}

int ksmi_pack_weights(const ksmi_pack_desc* d, int dtype, void* stream) {
  if (!d || d->nchunks < 1 || (d->nchunks > KSMI_MAX_CHUNKS && !d->uniform_kc)) return ksmi_fail(KSMI_E_ARG, "pack: bad descriptor");
  const int kc = ksmi_chunk_elems(dtype);
  const size_t total = (size_t)d->nchunks * d->taps * d->Npad * kc;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  if (dtype == KSMI_BF16) hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
  else hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *d);
  return ksmi_check_launch("pack_weights");
}

int ksmi_pack_weights_batched(const ksmi_pack_desc* descs_device, int n, int dtype, void* stream) {
  if (!descs_device || n < 1) return ksmi_fail(KSMI_E_ARG, "pack_batched: bad args");
  // x = workgroups per descriptor: the large tensors (512 x 512 x 9: 295 k vectors) decide the duration, so they get enough to fill
  // the machine on their own; workgroups beyond a small tensor's vectors fall through the loop at once (48 per descriptor: 108 us
  // per SNUNet step with ~200 workgroups alive in the tail)
  static const int gx = ksmi_knob_int("KSMI_PACK_GRID", 512);
  const dim3 grid(gx < 1 ? 1 : gx, n);
  if (dtype == KSMI_BF16) hipLaunchKernelGGL(pack_weights_batched_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, descs_device);
  else if (dtype == KSMI_F32) hipLaunchKernelGGL(pack_weights_batched_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, descs_device);
  else return ksmi_fail(KSMI_E_ARG, "pack_batched: bad dtype");
  return ksmi_check_launch("pack_weights_batched");
}

size_t ksmi_conv_wgrad_workspace(const ksmi_wgrad_desc* d, int dtype) {
  WgradGeom g = dtype == KSMI_BF16 ? wgrad_geom<bf16_t>(d) : wgrad_geom<float>(d);
  // caller reads nsplit back through the returned size: bytes = nsplit * slab
  return (size_t)g.nsplit * g.taps * d->nchunks * g.kc * g.npad * sizeof(float);
}

// 1: ksmi_conv_wgrad(d) also writes d->bias_grad (the token-GEMM path of gemm2.hip in its one-split, direct-write mode)
// 2 (round 5): the split (slab) mode of the same path writes the PARTIAL column sums of dY, one row per split, to d->bias_grad
//    [d->nsplit][d->N] floats (overwritten, not accumulated); the caller sums the rows into the bias gradient
int ksmi_conv_wgrad_fuses_bias(const ksmi_wgrad_desc* d, int dtype) {
  if (!d || dtype != KSMI_BF16 || d->nsrc != 1) return 0;
  static const bool off = ksmi_knob_is_set("KSMI_NO_FUSED_BIAS_GRAD");
  if (off) return 0;
  WgradGeom g = wgrad_geom<bf16_t>(d);
  if (!g.tn || g.v3) return 0;
  bool plain = d->gK == 1 && (!d->use_tap_off || d->tap_off[0] == 0) && d->gN >= d->src[0].c_len && d->gN < ((int64_t)1 << 31);
  if (plain && !d->uniform_kc)
    for (int i = 0; i < d->nchunks; ++i) plain = plain && d->k_off[i] == i * g.kc;
  if (!(plain && ksmi_gemm2_tn_enabled(d->src[0].c_len, d->N, g.rps))) return 0;
  if (g.nsplit == 1) return 1;
  return wgrad_slab_bias_on() && (g.tbt == 64 || g.tbt == 96 || g.tbt == 128) ? 2 : 0;
}

int ksmi_conv_wgrad(const ksmi_wgrad_desc* d, int dtype, void* stream) {
  if (!d || d->nsrc < 1 || d->nsrc > KSMI_MAX_SRC || d->nchunks < 1 || (d->nchunks > KSMI_MAX_CHUNKS && !(d->uniform_kc && d->nsrc == 1)))
    return ksmi_fail(KSMI_E_ARG, "wgrad: bad descriptor");
  if (dtype == KSMI_BF16) return launch_wgrad<bf16_t>(d, (hipStream_t)stream);
  if (dtype == KSMI_F32) return launch_wgrad<float>(d, (hipStream_t)stream);
  return ksmi_fail(KSMI_E_ARG, "wgrad: bad dtype");
}

}  // extern "C"
