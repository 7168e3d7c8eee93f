// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions (bf16), "channel-owner" tiling.
//
//   G[tap][k][n] = sum_pixels X[pixel + tap - 1][k] * dY[pixel][n]        (models/snunet.py:15-17 backward)
//
// A workgroup owns an output tile of CT = 16*WC input channels x NTL = 16*WN*NF output channels x ALL 9 taps and walks a
// contiguous range of 128-pixel patches (one "split" of the pixel axis).  Wave w owns ONE 16-channel fragment (cfi = w % WC)
// for every tap and NF column fragments: 9*NF accumulator tiles per wave, so a k-step of 32 pixels issues 9*NF MFMAs for
// 9 + NF fragment reads (the first kernel, igemm.hip: 12 MFMAs for 8 reads) and both operands of a patch are fetched from
// L2 / HBM once per CT x NTL tile instead of once per 32 x 32 tile.
// Both LDS images are plane-major ([32-channel plane][pixel row][64 B]), double-buffered and filled by LDS-DMA
// (global_load_lds_dwordx4: lane-linear destination, so the 32-byte bank swizzle is applied on the SOURCE address,
// cdna_hip_programming.md rule 21); positions outside the image are written as zeros with ds_write.  The MFMA K axis is the
// pixel axis: both operands are read transposed (ds_read_b64_tr_b16); a lane's fragment addresses are a table for k-step 0
// plus a uniform stride per k-step (the patch width divides 32, so a lane keeps its column and swizzle term in every k-step).
// A fused BN-apply + ReLU operand (conv2 of conv_block_nested reads relu(bn1(i)), snunet.py:24-25) is transformed IN LDS after
// the DMA has landed.  Partial slabs [split][tap][K][Npad] are summed by wgrad_reduce_kernel (igemm.hip), unchanged.
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"
#include "igemm_epilogue.h"
#include "wgrad3.h"
#include "dma.h"

namespace {

struct Wg3Args {
  ksmi_wgrad_desc d;
  int TH, TW, HWc, HPv;            // patch; staged halo width (min(TW, W) + 2); staged halo rows (TH + 2) * HWc
  int xpl, stage;                  // bytes of one X plane / of one stage
  int tilesX, tilesY, patches, pps;
  int KT, NTt;
  uint32_t m_hw, m_tw, m_tx, m_ty, m_tiles, m_ntt;
  int kstride;                     // bytes between k-steps inside an X plane
  int hymask;                      // 1: the swizzle term of the X image includes the halo row parity (TW == 8)
  int lds_bytes;
  int nst;                         // stages of the LDS ring (2 .. 4): patches i+1 .. i+nst-1 are in flight during the MFMAs of patch i
  const unsigned char* zero;       // >= 16 zero bytes (positions outside the image / the tile read it: every wave issues a FIXED
                                   // number of DMA instructions per stage, so the ring is waited for with counted vmcnt, dma.h)
};

__device__ __attribute__((aligned(64))) unsigned char wg3_zero_page[64];     // zero-initialised device memory
KSMI_DEVICE_SYMBOL_GETTER(wg3_zero, wg3_zero_page)

// LDS-DMA issued from inline asm (dma.h): hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 that
// follows a __builtin_amdgcn_global_load_lds (the transposed-read intrinsic carries no memory operand, so it may alias the DMA),
// which serialises the DMA of the next patches with the MFMAs of this one.  An asm DMA is invisible to that pass; its completion
// is waited for by hand (vm_wait with the per-wave instruction count of the stages issued later) in front of the barrier that
// publishes the stage.
__device__ __forceinline__ u32x4 tr_frag(unsigned a0, unsigned a1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
  return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// chunk tables as dword scalar loads (see igemm_epilogue.h: chunk_c0_of)
__device__ __forceinline__ int wg_chunk_c0(const ksmi_wgrad_desc& d, int ch) {
  if (d.uniform_kc) return ch * d.uniform_kc;
  const uint32_t w = ((const uint32_t*)d.chunk_c0)[ch >> 1];
  return (ch & 1) ? (int)(w >> 16) : (int)(w & 0xffffu);
}
__device__ __forceinline__ int wg_chunk_src(const ksmi_wgrad_desc& d, int ch) {
  if (d.uniform_kc) return 0;
  return (int)((((const uint32_t*)d.chunk_src)[ch >> 2] >> ((ch & 3) * 8)) & 0xffu);
}

// R0, C0 >= 0 (round 5): only the 2 x 2 window of taps (R0 .. R0+1) x (C0 .. C0+1) of the 3 x 3 neighbourhood is computed -- the weight
// gradient of one 2 x 2 PHASE convolution of ConvTranspose2d(k4, s2, p1) (plan_base._deconv_wgrad: X = a parity sub-image of d out as a
// strided view, padding p = 0 | 1 per axis <-> window origin 1 - p); slabs hold 4 taps, the reducer scatters through d.tap_off.
// PART (round 5): partial chunks (klen).  A template parameter: the run-time form cost the whole-chunk SNUNet launches 6-7 % (same box,
// single stream: 88 -> 93 us, 202 -> 217 us), so those compile exactly as before.
template <int WC, int WN, int NF, bool AFF, bool DEEP, int R0 = -1, int C0 = -1, bool PART = false>
__global__ __launch_bounds__(256, 2) void wgrad3_kernel(const Wg3Args ka) {
  typedef bf16_t T;
  const ksmi_wgrad_desc& d = ka.d;
  constexpr bool SUB = R0 >= 0;
  constexpr int NTAP = SUB ? 4 : 9;                // taps computed (and slabs per split)
  static_assert(!SUB || (R0 <= 1 && C0 >= 0 && C0 <= 1), "2 x 2 window inside the 3 x 3 neighbourhood");
  constexpr int CPT = WC / 2;                      // 32-channel planes of X per tile
  constexpr int NFT = WN * NF;                     // column fragments per tile
  constexpr int NPL = (NFT + 1) / 2;               // 32-column planes of dY per tile
  constexpr int NTL = NFT * 16;
  constexpr int XIT = 4;                           // X slot iterations: 256 * 4 slots of 16 B >= 4 * halo rows (<= 256)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cfi = wave % WC, wn = wave / WC;
  const int g = lane >> 4, l15 = lane & 15;

  // ---- which tile / split (XCD-aware: the tiles of one split are consecutive logical ids on one XCD and share its L2) ----
  const int L = (int)xcd_remap(blockIdx.x, gridDim.x);
  const FastDiv dTiles(ka.KT * ka.NTt, ka.m_tiles), dNTt(ka.NTt, ka.m_ntt);
  const int split = dTiles.div(L);
  const int rem = L - split * (ka.KT * ka.NTt);
  const int kt = dNTt.div(rem), nt = rem - kt * ka.NTt;
  const int n0 = nt * NTL;
  const int TW = ka.TW, TH = ka.TH, HWc = ka.HWc;
  const int P = TH * TW;
  const FastDiv dHW(HWc, ka.m_hw), dTW(TW, ka.m_tw), dTX(ka.tilesX, ka.m_tx), dTY(ka.tilesY, ka.m_ty);
  const int xpl = ka.xpl, stage = ka.stage;

  // ---- the tile's X planes ------------------------------------------------------------------------------------
  const unsigned char* sp[CPT];
  uint32_t cb[CPT];
  bool cvalid[CPT];
  int cc0[CPT], klen[CPT], krem[CPT];
  // klen: channels of the plane that exist in the tensor, rounded up to the 16-byte granule (32 except in the last chunk of a source
  // whose c_len is not a multiple of 32: the 16-channel levels of Unet / FC-Siam); the granules past it read the zero page, the
  // reducer skips the rows past k_len
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int ch = kt * CPT + c;
    cvalid[c] = ch < d.nchunks;
    const int chs = cvalid[c] ? ch : 0;
    const ksmi_src& sr = d.src[wg_chunk_src(d, chs)];
    cc0[c] = wg_chunk_c0(d, chs);
    sp[c] = (const unsigned char*)((const T*)sr.ptr + sr.c_off + cc0[c]);
    cb[c] = (uint32_t)sr.C * 2u;
    krem[c] = PART ? sr.c_len - cc0[c] : 32;
    klen[c] = krem[c] >= 32 ? 32 : ((krem[c] + 7) & ~7);
  }
  const unsigned char* const dyp = (const unsigned char*)((const T*)d.dy + d.dy_c_off);
  const uint32_t dycb = (uint32_t)d.dyC * 2u;

  // ---- zero the whole LDS allocation once (rows of the k padding, slack rows: everything an MFMA may touch is finite) ----
  for (int i = tid * 16; i < ka.lds_bytes; i += 256 * 16) *(u32x4*)(smem + i) = (u32x4){0u, 0u, 0u, 0u};
  float* aff_tab = (float*)(smem + ka.nst * stage);                      // AFF: [CPT*32][2] scale, shift of the tile's channels
  if constexpr (AFF) {
    __syncthreads();
    if (tid < CPT * 32) {
      const int c = tid >> 5, j = tid & 31;
      const ksmi_src& sr = d.src[0];
      const bool ok = cvalid[c] && (!PART || j < krem[c]);
      aff_tab[tid * 2 + 0] = ok ? sr.scale[cc0[c] + j] : 0.f;
      aff_tab[tid * 2 + 1] = ok ? sr.shift[cc0[c] + j] : 0.f;
    }
  }
  const bool aff_relu = AFF && d.src[0].relu != 0;

  // ---- fragment address tables of k-step 0, stage 0 --------------------------------------------------------------
  // X: [halo-row parity of the tap][tap column][half]; tap row 2 = tap row 0 + two halo rows (same parity, same swizzle term)
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned a_tab[2][3][2], b_tab[NF][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int p0 = g * 8 + h * 4 + (l15 >> 2);                      // pixel row of this lane's transposed read
    const int ly = dTW.div(p0), lx = p0 - ly * TW;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int hy = ly + ty, hx = lx + tx;
        const int f = ((hx >> 3) ^ (hy & ka.hymask)) & 1;
        a_tab[ty][tx][h] = lds0 + (unsigned)((cfi >> 1) * xpl + (hy * HWc + hx) * 64 + ((((cfi & 1) ^ f)) << 5) + (l15 & 3) * 8);
      }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int nfg = wn * NF + nf;
      const int fd = (p0 >> 3) & 1;
      b_tab[nf][h] = lds0 + (unsigned)(CPT * xpl + (nfg >> 1) * 8192 + p0 * 64 + ((((nfg & 1) ^ fd)) << 5) + (l15 & 3) * 8);
    }
  }
  const unsigned row2 = (unsigned)(2 * HWc * 64);

  f32x4 acc[NTAP][NF];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[t][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging ------------------------------------------------------------------------------------------------------
  auto patch_origin = [&](int patch, int& b, int& oy0, int& ox0) {
    const int q1 = dTX.div(patch);
    const int tx = patch - q1 * ka.tilesX;
    b = dTY.div(q1);
    const int ty = q1 - b * ka.tilesY;
    oy0 = ty * TH; ox0 = tx * TW;
  };
  // One stage = the X halo planes + the dY planes of a patch.  Every wave issues the SAME instructions for every patch: a slot
  // outside the image (or a column beyond N) reads the zero page, lanes beyond the halo are exec-masked but never a whole
  // instruction (the `wave_in` test is wave-uniform), so `dma_cnt` instructions per stage and wave is exact.
  const uint32_t zlo = (uint32_t)(uintptr_t)ka.zero, zhi = (uint32_t)((uint64_t)(uintptr_t)ka.zero >> 32);
  int nvalid = 0;
#pragma unroll
  for (int c = 0; c < CPT; ++c) nvalid += cvalid[c] ? 1 : 0;
  int dma_cnt = 2 * NPL;
#pragma unroll
  for (int it = 0; it < XIT; ++it) dma_cnt += (it * 256 + wave * 64 < ka.HPv * 4) ? nvalid : 0;
  dma_cnt = __builtin_amdgcn_readfirstlane(dma_cnt);
  auto issue_loads = [&](int patch, int stg) {
    int b, oy0, ox0;
    patch_origin(patch, b, oy0, ox0);
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int v = it * 256 + tid;
      const int R = v >> 2, s = v & 3;
      const int hy = dHW.div(R), hx = R - hy * HWc;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool inr = v < ka.HPv * 4;
      const bool ok = inr && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
      const int f = ((hx >> 3) ^ (hy & ka.hymask)) & 1;
      // (strided view: the phase gradients only -- compile-time, the dense 3 x 3 instances keep the round-4 address arithmetic)
      const uint32_t pix = (!SUB || d.in_sy == 0) ? (uint32_t)((b * d.Hin + iy) * d.Win + ix)
                                                  : (uint32_t)((b * d.in_H + (iy * d.in_sy + d.in_oy)) * d.in_W + (ix * d.in_sx + d.in_ox));
      const int sgr = s ^ (f << 1);                                    // source granule of this slot
      const uint32_t slb = (uint32_t)(sgr * 16);
      if (it * 256 + wave * 64 < ka.HPv * 4) {                       // wave-uniform: this wave has halo slots in this iteration
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          if (!cvalid[c]) continue;
          const bool okc = PART ? (ok && sgr * 8 < klen[c]) : ok;
          const uint64_t av = (uint64_t)(uintptr_t)sp[c] + (uint64_t)(pix * cb[c] + slb);
          const uint32_t lo = okc ? (uint32_t)av : zlo, hi = okc ? (uint32_t)(av >> 32) : zhi;
          if (inr) glds16_flat((const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo),
                               lds0 + (unsigned)(stg * stage + c * xpl + (it * 256 + wave * 64) * 16));
        }
      }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int v = it * 256 + tid;
      const int p = v >> 2, s = v & 3;
      const int ly = dTW.div(p), lx = p - ly * TW;
      const bool pv = p < P && oy0 + ly < d.Hout && ox0 + lx < d.Wout;
      const uint32_t pix = (uint32_t)((b * d.Hout + oy0 + ly) * d.Wout + ox0 + lx);
      const int sl = s ^ (((p >> 3) & 1) << 1);
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int n = n0 + j * 32 + sl * 8;
        const bool okn = pv && n < d.N;
        const uint64_t av = (uint64_t)(uintptr_t)dyp + (uint64_t)(pix * dycb + (uint32_t)(n * 2));
        const uint32_t lo = okn ? (uint32_t)av : zlo, hi = okn ? (uint32_t)(av >> 32) : zhi;
        glds16_flat((const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo),
                    lds0 + (unsigned)(stg * stage + CPT * xpl + j * 8192 + (it * 256 + wave * 64) * 16));
      }
    }
  };
  // AFF: relu(x * scale + shift) over the in-image positions of the landed X planes (padding stays zero: it applies AFTER the transform)
  auto transform = [&](int patch, int stg) {
    int b, oy0, ox0;
    patch_origin(patch, b, oy0, ox0);
    unsigned char* const sb = smem + stg * stage;
    const int sl = tid & 3;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (!cvalid[c]) continue;
      float sc[8], sh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { sc[j] = aff_tab[(c * 32 + sl * 8 + j) * 2]; sh[j] = aff_tab[(c * 32 + sl * 8 + j) * 2 + 1]; }
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        const int v = it * 256 + tid;
        const int R = v >> 2;
        const int hy = dHW.div(R), hx = R - hy * HWc;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        if (v < ka.HPv * 4 && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win) {
          const int f = ((hx >> 3) ^ (hy & ka.hymask)) & 1;
          u32x4* q = (u32x4*)(sb + c * xpl + R * 64 + ((sl ^ (f << 1)) * 16));
          float x[8];
          vec_unpack<T>(*q, x);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            x[j] = x[j] * sc[j] + sh[j];
            if (aff_relu) x[j] = fmaxf(x[j], 0.f);
          }
          *q = vec_pack<T>(x);
        }
      }
    }
  };

  const int p_begin = split * ka.pps;
  const int p_end = min(ka.patches, p_begin + ka.pps);
  const int npat = p_end - p_begin;
  const int NST = ka.nst;
  const bool mine = cvalid[cfi >> 1];                               // this wave's channel fragment exists
  __syncthreads();                                                  // zero fill (and the affine table) visible; no DMA is in flight yet
  for (int j = 0; j < NST - 1 && j < npat; ++j) issue_loads(p_begin + j, j);
  int stg = 0;
  for (int i = 0; i < npat; ++i) {
    const int patch = p_begin + i;
    const int ahead = min(NST - 2, npat - 1 - i);                   // stages issued after this patch's
    vm_wait(ahead * dma_cnt);                                       // this wave's DMA of stage `stg` has landed ...
    lds_barrier();                                                  // ... and everybody's; the stage of patch i-1 is free (its MFMAs are done)
    if (i + NST - 1 < npat) issue_loads(patch + NST - 1, stg == 0 ? NST - 1 : stg - 1);
    if constexpr (AFF) {
      transform(patch, stg);
      lds_barrier();
    }
    const unsigned so = (unsigned)(stg * stage);
    stg = stg + 1 == NST ? 0 : stg + 1;
    if (!mine) continue;
    // flat (k-step, tap) sequence q = 9 ks + t.  A wave that is alone on its SIMD gets its transposed fragments back after the
    // LDS latency (MI355X_MICROARCH.md, LDS: one wave reaches a fifth of the ds_read_b64 rate), and one tap is only NF MFMAs
    // (16 cycles each): the X fragment of tap q + AD is requested right after the MFMAs of tap q have issued (ring of AD
    // fragments), the dY fragments of the next k-step during the second half of this one (two sets).
    constexpr int AD = !DEEP ? 1 : NF >= 4 ? 4 : (NF == 2 ? 6 : 9);   // (DEEP = false: the one-tap schedule of rounds 2-4, kept for A/B)
    auto a_frag = [&](int q) {
      const int ks = q / NTAP, t = q % NTAP, ty = SUB ? R0 + t / 2 : t / 3, tx = SUB ? C0 + t % 2 : t % 3;
      const unsigned o = so + (unsigned)(ks * ka.kstride) + (ty == 2 ? row2 : 0u);
      return tr_frag(a_tab[ty & 1][tx][0] + o, a_tab[ty & 1][tx][1] + o);
    };
    u32x4 af[AD], bf[2][NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bf[0][nf] = tr_frag(b_tab[nf][0] + so, b_tab[nf][1] + so);
#pragma unroll
    for (int q = 0; q < AD; ++q) af[q] = a_frag(q);
    if constexpr (DEEP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int q = ks * NTAP + t;
        if (t == NTAP / 2 && ks + 1 < 4) {
          const unsigned bo = so + (unsigned)((ks + 1) * 2048);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) bf[(ks + 1) & 1][nf] = tr_frag(b_tab[nf][0] + bo, b_tab[nf][1] + bo);
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) mma16<T>(acc[t][nf], af[q % AD], bf[ks & 1][nf]);
        if (q + AD < 4 * NTAP) af[q % AD] = a_frag(q + AD);
        if constexpr (DEEP) __builtin_amdgcn_sched_barrier(0);        // (the scheduler sinks the requests back to one tap ahead otherwise)
      }
    }
  }
  if (!mine) return;
  // ---- partial slab [split][tap][K][Npad]: row = channel (g*4 + r of the fragment), column = n (l15) ----------------
  const int Npad = (d.N + 15) & ~15;
  const int Ktot = d.nchunks * 32;
  const int krow0 = (kt * CPT + (cfi >> 1)) * 32 + (cfi & 1) * 16 + g * 4;
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int n = n0 + (wn * NF + nf) * 16 + l15;
      if (n >= Npad) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        d.partial[(((size_t)split * NTAP + t) * Ktot + krow0 + r) * Npad + n] = acc[t][nf][r];
    }
}

// Sum of the split slabs + scatter into the fp32 gradient (fixed summation tree => deterministic).  A block owns 16 float4
// (256 contiguous bytes of a slab) x 16 slab lanes: every wave instruction reads four 256-byte runs; slab lane sl walks slabs
// sl, sl + 16, ...; the 16 lanes of an element are combined by two shuffles (inside a wave) and one LDS step (across the waves).
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const ksmi_wgrad_desc d, int nsplit, int ntaps) {
  __shared__ f32x4 red[4][16];
  const int Npad = (d.N + 15) & ~15;
  const int Ktot = d.nchunks * 32;
  const size_t total4 = (size_t)ntaps * Ktot * Npad / 4;
  const int tid = threadIdx.x, e = tid & 15, sl = tid >> 4;
  const size_t i4 = (size_t)blockIdx.x * 16 + e;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
    const f32x4* p = (const f32x4*)d.partial + i4;
    int sp = sl;
    for (; sp + 16 < nsplit; sp += 32) { a0 += p[(size_t)sp * total4]; a1 += p[(size_t)(sp + 16) * total4]; }
    if (sp < nsplit) a0 += p[(size_t)sp * total4];
  }
  f32x4 s = a0 + a1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s[j] += __shfl_xor(s[j], 16, 64);
    s[j] += __shfl_xor(s[j], 32, 64);
  }
  if ((tid & 63) < 16) red[tid >> 6][e] = s;
  __syncthreads();
  if (tid < 16 && i4 < total4) {
    const f32x4 v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const size_t i = i4 * 4;
    const int n = (int)(i % Npad);
    const size_t r = i / Npad;
    const int krow = (int)(r % Ktot), t = (int)(r / Ktot);
    const int ch = krow >> 5;
    const int64_t k = (d.uniform_kc ? ch * d.uniform_kc : d.k_off[ch]) + (krow & 31);
    const int kl = d.uniform_kc ? min(d.uniform_kc, d.k_total - ch * d.uniform_kc) : d.k_len[ch];   // (rows past it: pad channels of a partial chunk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n + j < d.N && (krow & 31) < kl) {
        float* gp = d.grad + k * d.gK + (int64_t)(n + j) * d.gN + (d.use_tap_off ? (int64_t)d.tap_off[t] : (int64_t)t * d.gT);
        *gp = d.accumulate ? (*gp + v[j]) : v[j];
      }
    }
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------
static bool partial_on_n() {     // KSMI_WGRAD3_PARTIAL=0: whole chunks, N % 8 == 0, N >= 16 only (the round-4 rule; same-box A/B)
  static const bool on = (ksmi_knob_int("KSMI_WGRAD3_PARTIAL", 1) != 0);
  return on;
}
bool ksmi_wgrad3_geom(const ksmi_wgrad_desc* d, int dtype, ksmi_wgrad3_geom_t* g) {
  static const bool off = ksmi_knob_is_set("KSMI_WGRAD3_OFF");
  if (off || dtype != KSMI_BF16) return false;
  // 2 x 2 phase gradients (ConvTranspose2d(k4, s2, p1), plan_base._deconv_wgrad): the 2 x 2 window of the 3 x 3 neighbourhood whose
  // origin is (1 - pad, 1 - pad_x); KSMI_WGRAD3_K2=0 sends them back to igemm_wgrad_kernel
  static const int k2_on = ksmi_knob_int("KSMI_WGRAD3_K2", 1);
  const bool k2 = d->KH == 2 && d->KW == 2;
  g->r0 = g->c0 = -1;
  if (k2) {
    if (!k2_on || d->stride != 1 || (unsigned)d->pad > 1u || !d->pad_x_set || (unsigned)d->pad_x > 1u || !d->use_tap_off || d->src[0].scale) return false;
    g->r0 = 1 - d->pad; g->c0 = 1 - d->pad_x;
  } else if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || (d->pad_x_set && d->pad_x != 1) || d->in_sy != 0 || d->use_tap_off) return false;
  if (d->Hin != d->Hout || d->Win != d->Wout || d->N < 1) return false;
  // N < 16 or N % 8 != 0 (the weight gradient of a 2- / 3-class head: d out has a channel stride of 8): whole 16-byte granules of the
  // d out row are read (the pad channels must exist inside the row; their columns are computed and dropped by the reducer)
  if (((d->N % 8) != 0 || d->N < 16) && (!partial_on_n() || k2 || (d->dyC % 8) || (d->dy_c_off % 8) || d->dy_c_off + ((d->N + 7) & ~7) > d->dyC)) return false;
  // KSMI_WGRAD3_PARTIAL=0: whole 32-channel chunks only (the round-4 rule; same-box A/B)
  static const bool partial_on = (ksmi_knob_int("KSMI_WGRAD3_PARTIAL", 1) != 0);
  for (int i = 0; i < d->nsrc; ++i) {
    // whole 16-byte granules inside the source row; the last chunk of a source may be partial (zero-page granules, klen in the kernel)
    if (d->src[i].c_len < 1 || (d->src[i].C % 8) || (d->src[i].c_off % 8) || d->src[i].c_off + ((d->src[i].c_len + 7) & ~7) > d->src[i].C) return false;
    if ((d->src[i].c_len % 32) && (!partial_on || k2)) return false;
    if (d->src[i].scale && (i > 0 || d->nsrc != 1)) return false;
    const size_t ipx = d->in_sy ? (size_t)d->B * d->in_H * d->in_W : (size_t)d->B * d->Hin * d->Win;
    if (ipx * (size_t)d->src[i].C * 2 >= ((size_t)1 << 32)) return false;
  }
  if ((size_t)d->B * d->Hout * d->Wout * (size_t)d->dyC * 2 >= ((size_t)1 << 32)) return false;
  if (d->uniform_kc) { if (d->uniform_kc != 32 || d->k_total > d->nchunks * 32 || d->k_total <= (d->nchunks - 1) * 32) return false; }
  else for (int i = 0; i < d->nchunks; ++i) if (d->k_len[i] < 1 || d->k_len[i] > 32 || (d->k_len[i] != 32 && (!partial_on || k2))) return false;
  // patch: width 8 / 16 / 32, at most 128 pixels; fewest patches, then smallest halo
  const int H = d->Hout, W = d->Wout;
  long best = -1;
  for (int tw = 8; tw <= 32; tw *= 2) {
    const int thmax = 128 / tw;
    const int tilesX = (W + tw - 1) / tw;
    const int tilesY0 = (H + thmax - 1) / thmax;
    const int th = (H + tilesY0 - 1) / tilesY0;                    // smallest patch height with the same number of patch rows
    const int hwc = (tw < W ? tw : W) + 2;
    const long key = (long)tilesX * tilesY0 * 100000 + (long)(thmax + 2) * hwc;
    if ((thmax + 2) * hwc > 256) continue;
    if (best < 0 || key <= best) { best = key;   // ties: the wider patch (longer contiguous rows)
      g->TW = tw; g->TH = th; g->HWc = hwc; g->tilesX = tilesX; g->tilesY = tilesY0; }
  }
  if (best < 0) return false;
  const int npad = (d->N + 15) & ~15;
  // tile: one chunk -> 32 channels (2 x 2 waves), else 64 channels (4 x 1 waves); 64 columns when there are at least 48
  g->WC = d->nchunks == 1 ? 2 : 4;
  g->NTL = npad >= 48 ? 64 : 32;
  if (k2 && (g->WC != 4 || g->NTL != 64)) return false;            // (one instance per window: 64 channels x 64 columns)
  g->KT = (d->nchunks + g->WC / 2 - 1) / (g->WC / 2);
  g->NTt = (npad + g->NTL - 1) / g->NTL;
  g->patches = d->B * g->tilesX * g->tilesY;
  const int hh_alloc = 128 / g->TW + 2;
  g->xpl = hh_alloc * g->HWc * 64;
  g->stage = (g->WC / 2) * g->xpl + (g->NTL / 32) * 8192;
  if (2 * (size_t)g->stage + 1024 > 160 * 1024) return false;
  // LDS ring: KSMI_WGRAD3_NST = 3 / 4 lets a workgroup that owns its CU (at most 256 workgroups) keep up to that many stages
  // (patches i+1 .. i+nst-1 in flight during the MFMAs of patch i); the default is two stages and two workgroups per CU.
  const int nst_knob = ksmi_knob_int("KSMI_WGRAD3_NST", 2);         // (a knob: the tests force each depth)
  const int nst_want = nst_knob >= 2 ? nst_knob : 2;
  int nst_max = (int)((160 * 1024 - 1024) / g->stage);
  if (nst_max > 4) nst_max = 4;
  if (nst_want < nst_max) nst_max = nst_want;
  // number of workgroups: every workgroup dumps its 9 x CT x NTL fp32 tile once, so small gradients pay for many splits in slab
  // traffic (K = N = 64 at 512 workgroups: 75 MB of slabs against 103 MB of operands), while too few workgroups cannot keep
  // HBM busy.  Pick the count that minimises, in microseconds with coarse measured constants,
  //   max(MFMA time, operand bytes / min(4.5 TB/s, workgroups x stages in flight / 5 us)) + slab write + slab read.
  const int wg_force = ksmi_knob_int("KSMI_WGRAD3_WGS", 0);         // (a knob: the tests shrink the grid so that the ring wraps)
  const int tiles = g->KT * g->NTt;
  const int nfw = (g->NTL / 16) / (4 / g->WC);                      // column fragments per wave
  const double ntaps = k2 ? 4.0 : 9.0;
  const double t_patch = 4.0 * ntaps * nfw * 16.0 / 2100.0;         // one 128-pixel patch on one wave at the MFMA rate
  const double tile_mb = ntaps * (g->WC * 16) * g->NTL * 4.0 / 1e6;
  const double op_bytes = (double)d->B * H * W * (d->nchunks * 32.0 + npad) * 2.0;
  const size_t lds2 = 2 * (size_t)g->stage + 1024;
  const int max_per_cu = (int)(160 * 1024 / lds2) < 4 ? (int)(160 * 1024 / lds2) : 4;
  // MFMA-phase efficiency by workgroups per CU.  One workgroup per CU = one wave per SIMD: nothing covers its barriers, DMA issue and
  // operand transform (measured 3.9 us per patch against 1.1 us of MFMAs: profiles/r04_wgrad3_wgs.txt, K = N = 128 at 56^2:
  // 76 us on 256 workgroups, 66 us on 512 in spite of twice the slab traffic)
  static const double eff[5] = {0.0, 0.35, 0.8, 0.85, 0.9};
  double best_t = 1e30;
  static const int cand[] = {128, 192, 256, 384, 512, 768, 1024};
  for (int ci = 0; ci < 7; ++ci) {
    const int wg = wg_force ? wg_force : cand[ci];
    if (wg > 256 * max_per_cu && !wg_force) break;
    int want = wg / tiles;
    if (want < 1) want = 1;
    if (want > g->patches) want = g->patches;
    const int pps = (g->patches + want - 1) / want;
    const int ns = (g->patches + pps - 1) / pps;
    int per_cu = (ns * tiles + 255) / 256;
    if (per_cu > 4) per_cu = 4;
    const int nst = per_cu == 1 ? nst_max : 2;
    const double e = eff[per_cu];
    double t_main = pps * t_patch * per_cu / e;
    double bw = (double)ns * tiles * g->stage / 5.0;                 // bytes per microsecond in flight (one stage per workgroup: a deeper
                                                                     // ring measured no faster, and the split count must not depend on it)
    if (bw > 4.5e6) bw = 4.5e6;
    if (t_main < op_bytes / bw) t_main = op_bytes / bw;
    const double t = t_main + 7.0 + 2.0 * ns * tiles * tile_mb / 4.0;
    if (t < best_t) { best_t = t; g->pps = pps; g->nsplit = ns; g->nst = nst; }
  }
  g->lds = (size_t)g->nst * g->stage + 1024;                        // slack: AFF table
  return true;
}

int ksmi_wgrad3_reduce(const ksmi_wgrad_desc* d, const ksmi_wgrad3_geom_t* g, hipStream_t st) {
  const int ntaps = g->r0 >= 0 ? 4 : 9;
  const size_t total4 = (size_t)ntaps * d->nchunks * 32 * ((d->N + 15) & ~15) / 4;
  hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3((unsigned)((total4 + 15) / 16)), dim3(256), 0, st, *d, g->nsplit, ntaps);
  return ksmi_check_launch("wgrad3_reduce");
}

int ksmi_wgrad3_launch(const ksmi_wgrad_desc* d, const ksmi_wgrad3_geom_t* g, hipStream_t st) {
  Wg3Args ka;
  ka.d = *d;
  ka.TH = g->TH; ka.TW = g->TW; ka.HWc = g->HWc; ka.HPv = (g->TH + 2) * g->HWc;
  ka.xpl = g->xpl; ka.stage = g->stage;
  ka.tilesX = g->tilesX; ka.tilesY = g->tilesY; ka.patches = g->patches; ka.pps = g->pps;
  ka.KT = g->KT; ka.NTt = g->NTt;
  ka.m_hw = fastdiv_magic(g->HWc); ka.m_tw = fastdiv_magic(g->TW); ka.m_tx = fastdiv_magic(g->tilesX); ka.m_ty = fastdiv_magic(g->tilesY);
  ka.m_tiles = fastdiv_magic(g->KT * g->NTt); ka.m_ntt = fastdiv_magic(g->NTt);
  ka.kstride = (32 / g->TW) * g->HWc * 64;
  ka.hymask = g->TW == 8 ? 1 : 0;
  ka.lds_bytes = (int)g->lds;
  ka.nst = g->nst;
  const unsigned char* const zero_page = wg3_zero();
  if (!zero_page) return ksmi_fail(KSMI_E_UNSUPPORTED, "wgrad3: zero page");
  ka.zero = zero_page;
  if ((size_t)g->patches * 1 >= ((size_t)1 << 31) / 1024) return ksmi_fail(KSMI_E_UNSUPPORTED, "wgrad3: too many patches");
  const dim3 grid(g->nsplit * g->KT * g->NTt);
  const bool aff = d->src[0].scale != nullptr;
  static const bool deep = (ksmi_knob_int("KSMI_WGRAD3_DEEP", 1) != 0);   // fragment requests AD taps ahead (default)
#define KSMI_W3_(WC_, WN_, NF_, AFF_, DEEP_)                                                          \
  do {                                                                                               \
    auto kfn = wgrad3_kernel<WC_, WN_, NF_, AFF_, DEEP_>; KSMI_NOTE(kfn);                            \
    if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256), g->lds, st, ka);                                        \
  } while (0)
#define KSMI_W3(WC_, WN_, NF_)                                                                       \
  do {                                                                                               \
    if (aff) { if (deep) KSMI_W3_(WC_, WN_, NF_, true, true); else KSMI_W3_(WC_, WN_, NF_, true, false); }   \
    else { if (deep) KSMI_W3_(WC_, WN_, NF_, false, true); else KSMI_W3_(WC_, WN_, NF_, false, false); }     \
  } while (0)
  if (g->r0 >= 0) {                                                 // 2 x 2 phase gradients: one instance per window origin
#define KSMI_W3K(R0_, C0_)                                                                           \
  do {                                                                                               \
    auto kfn = wgrad3_kernel<4, 1, 4, false, true, R0_, C0_>; KSMI_NOTE(kfn);                        \
    if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256), g->lds, st, ka);                                        \
  } while (0)
    if (aff || g->WC != 4 || g->NTL != 64) return ksmi_fail(KSMI_E_UNSUPPORTED, "wgrad3: no 2 x 2 instance");
    if (g->r0 == 0 && g->c0 == 0) KSMI_W3K(0, 0); else if (g->r0 == 0) KSMI_W3K(0, 1); else if (g->c0 == 0) KSMI_W3K(1, 0); else KSMI_W3K(1, 1);
#undef KSMI_W3K
    return ksmi_check_launch("wgrad3");
  }
  bool part = false;
  for (int i = 0; i < d->nsrc; ++i) part = part || (d->src[i].c_len % 32) != 0;
  if (part) {                                                       // partial chunks: their own instances (fragment requests ahead: on)
#define KSMI_W3P(WC_, WN_, NF_)                                                                      \
  do {                                                                                               \
    if (aff) { auto kfn = wgrad3_kernel<WC_, WN_, NF_, true, true, -1, -1, true>; KSMI_NOTE(kfn);    \
      if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
      hipLaunchKernelGGL(kfn, grid, dim3(256), g->lds, st, ka); }                                    \
    else { auto kfn = wgrad3_kernel<WC_, WN_, NF_, false, true, -1, -1, true>; KSMI_NOTE(kfn);       \
      if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
      hipLaunchKernelGGL(kfn, grid, dim3(256), g->lds, st, ka); }                                    \
  } while (0)
    if (g->WC == 4 && g->NTL == 64) KSMI_W3P(4, 1, 4);
    else if (g->WC == 4) KSMI_W3P(4, 1, 2);
    else if (g->NTL == 64) KSMI_W3P(2, 2, 2);
    else KSMI_W3P(2, 2, 1);
#undef KSMI_W3P
    return ksmi_check_launch("wgrad3");
  }
  if (g->WC == 4 && g->NTL == 64) KSMI_W3(4, 1, 4);
  else if (g->WC == 4) KSMI_W3(4, 1, 2);
  else if (g->NTL == 64) KSMI_W3(2, 2, 2);
  else KSMI_W3(2, 2, 1);
#undef KSMI_W3
#undef KSMI_W3_
  return ksmi_check_launch("wgrad3");
}
