// Implicit-GEMM convolution for SHORT K (<= 8 k-chunks x taps <= 18 weight fragments per column fragment), bf16:
// persistent workgroups with the weights held in REGISTERS.
//
// The layers this serves are bandwidth- or latency-bound, not MFMA-bound (SNUNet: conv2 / conv1 of the 32- and 64-channel levels,
// ConvTranspose2d k2 s2 forward as a 1x1 GEMM over 4C columns, its input gradient as a 2x2 stride-2 convolution): in igemm2.hip a
// workgroup of such a layer is one dependent chain (tables -> DMA wait -> 72..144 MFMAs -> epilogue) and re-fetches its weight slab
// per k-chunk.  Here a workgroup
//   * loads the weight fragments of its column tile ONCE into VGPRs (MFMA A operand, swapped-operand orientation of igemm2),
//   * walks pixel tiles t = blockIdx.x, blockIdx.x + gridDim.x, ... with the halo image of tile i+1 in flight (LDS-DMA, two
//     stages) while the MFMAs and the epilogue stores of tile i run,
//   * issues the SAME number of DMA instructions for every tile: positions outside the image read a zero page instead of being
//     exec-masked (zero padding without ds_write fills or per-tile zeroing),
//   * keeps BatchNorm statistics in registers across its tiles: `stats` has one row per workgroup (ksmi_conv_stats_rows) instead
//     of one per tile, so the finalize kernels need no folding pass.
// A fused BN-apply + ReLU operand (conv2 reads relu(bn1(i)), models/snunet.py:24-25) is transformed in LDS after landing.
// Epilogue = the lean epilogue of igemm_epilogue.h (bias, ReLU-mask + BN-backward sums, pixel shuffle), restated for persistence.
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"
#include "igemm_epilogue.h"
#include "igemm3.h"
#include "dma.h"

namespace {

__device__ __attribute__((aligned(64))) unsigned char ig3_zero_page[64];      // zero-initialised device memory
KSMI_DEVICE_SYMBOL_GETTER(ig3_zero, ig3_zero_page)

struct Ig3Args {
  ksmi_conv_desc d;
  uint32_t m_tw, m_hw, m_tx, m_ty;
  int tiles, hpb, nslot, stage;
  int dbg;                         // KSMI_IG3_DBG profiling switches (wrong results): 1 skip the AFF transform, 2 skip its barrier too, 4 skip the epilogue stores
  int ns;                          // LDS stages: 3 (two tiles in flight) when they fit, else 2
  const unsigned char* zero;       // >= 16 zero bytes (positions outside the image read them)
};

// DIR: 1 = input-gradient launch (ksmi_conv_desc.dir): a name tag for profilers, no code difference
// PART (round 5): partial chunks (klen), N % 8 != 0 and several destinations.  A template parameter, not run-time tests: with the extra
// live values in the tile loop the instances that hold 72-144 weight registers ran 6-24 % longer on the SNUNet shapes (same-box,
// single stream) -- the whole-chunk instances compile exactly as before
template <int KH, int KW, int NCH, int WN, bool AFF, int NTI, bool MASK, int DIR = 0, bool PART = false>
__global__ __launch_bounds__(256 * WN, (WN == 1 && NCH * KH * KW * NTI <= 9) ? 2 : 1) void igemm3_kernel(const Ig3Args ka) {
  typedef bf16_t T;
  const ksmi_conv_desc& d = ka.d;
  constexpr int NTHR = 256 * WN;
  constexpr int TAPS = KH * KW;
  constexpr int BN = 32 * WN;
  constexpr int MAXSLOT = 2048 / NTHR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int g = lane >> 4, l15 = lane & 15;
  const int S = d.stride;
  const int HH = (d.TH - 1) * S + KH, HW = (d.TW - 1) * S + KW;
  const int HP = HH * HW;
  const int P = d.TH * d.TW;
  const int tilesX = (d.Wout + d.TW - 1) / d.TW, tilesY = (d.Hout + d.TH - 1) / d.TH;
  const FastDiv dTX(tilesX, ka.m_tx), dTY(tilesY, ka.m_ty), dHW(HW, ka.m_hw), dTW(d.TW, ka.m_tw);
  const int hpb = ka.hpb, stage = ka.stage;

  // ---- tile-invariant tables ------------------------------------------------------------------------------------------
  // halo slot v = tid + s * NTHR <-> (pixel v >> 2, 16-byte k-group slot v & 3); (hy, hx) are recomputed where needed (one
  // multiply-high each: cheaper than 2 * MAXSLOT table registers next to 72-144 registers of weights)
  int a_base[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    int p = wm * 64 + mf * 16 + l15;
    if (p >= P) p = 0;
    const int ly = dTW.div(p), lx = p - ly * d.TW;
    a_base[mf] = ly * S * HW + lx * S;
  }
  // per-chunk source (virtual concat): base pointer of the chunk's first channel, bytes per pixel
  // klen: channels of the chunk that exist in the tensor, rounded up to the 16-byte granule (32 except in the last chunk of a source
  // whose c_len is not a multiple of 32 -- the 16-channel levels of Unet / FC-Siam, 8-channel-stride heads): the granules past it are
  // read from the zero page like the positions outside the image, and the packed weights hold zeros there
  const unsigned char* sp[NCH];
  uint32_t cb[NCH];
  int cc0[NCH], klen[NCH], krem[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const ksmi_src& sr = d.src[chunk_src_of(d, ch)];
    cc0[ch] = chunk_c0_of(d, ch);
    sp[ch] = (const unsigned char*)((const T*)sr.ptr + sr.c_off + cc0[ch]);
    cb[ch] = (uint32_t)sr.C * 2u;
    krem[ch] = PART ? sr.c_len - cc0[ch] : 32;
    klen[ch] = krem[ch] >= 32 ? 32 : ((krem[ch] + 7) & ~7);
  }
  // ---- weights -> registers: fragment nf, row j = l15 of the MFMA A operand = output channel 8*(j>>2) + 4*nf + (j&3) of the wave's
  //      32 columns (the lean epilogue's ownership: lane (g, l15) ends up with channels 8g .. 8g+7 of pixel l15) --------------------
  u32x4 wreg[NTI][NCH][TAPS][2];
#pragma unroll
  for (int ni = 0; ni < NTI; ++ni)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const int n = (blockIdx.y * NTI + ni) * BN + wn * 32 + 8 * (l15 >> 2) + 4 * nf + (l15 & 3);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          wreg[ni][ch][t][nf] = (u32x4){0u, 0u, 0u, 0u};
          if (n < d.Npad) wreg[ni][ch][t][nf] = *(const u32x4*)((const unsigned char*)d.wpk + ((size_t)(ch * TAPS + t) * d.Npad + n) * 64 + g * 16);
        }
    }
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  float* aff_tab = (float*)(smem + ka.ns * stage);                      // AFF: [NCH*32][2] scale, shift
  if constexpr (AFF) {
    if (tid < NCH * 32) {                                           // layout [chunk][k-group q]{scale[8], shift[8]}
      const int ch = tid >> 5, j = tid & 31;
      const ksmi_src& sr = d.src[0];
      const bool kv = !PART || j < krem[ch];                                   // (channels past the source: 0 * x + 0 on zero-page granules)
      aff_tab[(ch * 4 + (j >> 3)) * 16 + (j & 7)] = kv ? sr.scale[cc0[ch] + j] : 0.f;
      aff_tab[(ch * 4 + (j >> 3)) * 16 + 8 + (j & 7)] = kv ? sr.shift[cc0[ch] + j] : 0.f;
    }
  }
  const bool aff_relu = AFF && d.src[0].relu != 0;

  // ---- staging --------------------------------------------------------------------------------------------------------
  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int q1 = dTX.div(t);
    const int tx = t - q1 * tilesX;
    b = dTY.div(q1);
    const int ty = q1 - b * tilesY;
    oy0 = ty * d.TH; ox0 = tx * d.TW;
  };
  auto issue = [&](int t, int stg) {
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    const int iy0 = oy0 * S - d.pad, ix0 = ox0 * S - d.pad_x;
    int tv = tid;
    asm volatile("" : "+v"(tv));          // opaque per call: keeps hipcc from hoisting the (tile-invariant) slot arithmetic out of the
                                          // tile loop into 30+ table registers (the weights already hold 72-144)
#pragma unroll
    for (int s = 0; s < MAXSLOT; ++s) {
      if (s < ka.nslot) {
        const int v = tv + s * NTHR;
        const int pix = v >> 2;
        const int hy = dHW.div(pix), hx = pix - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = v < HP * 4 && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        const uint32_t gpix = (uint32_t)((b * d.Hin + iy) * d.Win + ix);
        const int gq = (v & 3) ^ swz(pix);
        const uint32_t qb = (uint32_t)(gq << 4);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          // (two 32-bit selects: a pointer select compiles to two exec-masked DMA instructions)
          const bool okc = PART ? (ok && gq * 8 < klen[ch]) : ok;
          const uint64_t av = (uint64_t)(uintptr_t)sp[ch] + (uint64_t)gpix * cb[ch] + (uint64_t)qb;
          const uint64_t zv = (uint64_t)(uintptr_t)ka.zero;
          const uint32_t lo = okc ? (uint32_t)av : (uint32_t)zv, hi = okc ? (uint32_t)(av >> 32) : (uint32_t)(zv >> 32);
          const unsigned char* src = (const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo);
          glds16_flat(src, lds0 + (unsigned)(stg * stage + ch * hpb + (s * NTHR + wave * 64) * 16));
        }
      }
    }
  };
  // AFF: relu(x * scale + shift) over the in-image positions of the landed halo images (padding stays zero).  Straight-line: all
  // reads of a chunk are issued before the first use (thread = k-group q of pixel (tid + s * NTHR) >> 2; the table row is lane-constant)
  auto transform = [&](int t, int stg) {
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    const int iy0 = oy0 * S - d.pad, ix0 = ox0 * S - d.pad_x;
    unsigned char* const sb = smem + stg * stage;
    int tv = tid;
    asm volatile("" : "+v"(tv));          // (see issue)
    const int q = tv & 3;
    bool okv[MAXSLOT];
#pragma unroll
    for (int s = 0; s < MAXSLOT; ++s) {
      const int v = tv + s * NTHR;
      const int pix = v >> 2;
      const int hy = dHW.div(pix), hx = pix - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      okv[s] = s < ka.nslot && v < HP * 4 && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const f32x4* tab = (const f32x4*)(aff_tab + (ch * 4 + q) * 16);       // [chunk][k-group]{scale[8], shift[8]}
      const f32x4 s0 = tab[0], s1 = tab[1], h0 = tab[2], h1 = tab[3];
      constexpr int BATCH = MAXSLOT > 4 ? 4 : MAXSLOT;               // reads in flight per batch (registers)
#pragma unroll
      for (int s0i = 0; s0i < MAXSLOT; s0i += BATCH) {
        u32x4 xv[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int pix = (tv + (s0i + k) * NTHR) >> 2;
          if (s0i + k < ka.nslot) xv[k] = *(const u32x4*)(sb + ch * hpb + pix * 64 + ((q ^ swz(pix)) << 4));
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          if (s0i + k < ka.nslot) {
            const int pix = (tv + (s0i + k) * NTHR) >> 2;
            float x[8];
            vec_unpack<T>(xv[k], x);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              x[j] = x[j] * s0[j] + h0[j];
              x[4 + j] = x[4 + j] * s1[j] + h1[j];
            }
            if (aff_relu) {             // one v_med3_f32 per value (fmaxf compiles to a canonicalising v_max plus the v_max)
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_fmed3f(x[j], 0.f, 3.0e38f);
            }
            u32x4 o = vec_pack<T>(x);
            if (ka.dbg & 16) o = xv[k];
            if (!okv[s0i + k]) o = (u32x4){0u, 0u, 0u, 0u};
            if (!(ka.dbg & 8)) *(u32x4*)(sb + ch * hpb + pix * 64 + ((q ^ swz(pix)) << 4)) = o;
            else asm volatile("" :: "v"(o));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- epilogue state ---------------------------------------------------------------------------------------------------
  constexpr bool has_mask = MASK;
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }

  // bias of the column tile(s): loaded once (a global load inside the tile loop would make hipcc drain the DMA queue for it)
  float biasr[NTI][8];
#pragma unroll
  for (int ni = 0; ni < NTI; ++ni) {
    const int nc = (blockIdx.y * NTI + ni) * BN + wn * 32 + g * 8;
    int nn = nc;
    if (d.ps_cout > 0) nn = nc - (nc / d.ps_cout) * d.ps_cout;
#pragma unroll
    for (int j = 0; j < 8; ++j) biasr[ni][j] = 0.f;
    if (d.bias && (PART ? nc + 8 <= d.N : nc < d.N)) {
      const f32x4 a = *(const f32x4*)(d.bias + nn), c = *(const f32x4*)(d.bias + nn + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { biasr[ni][j] = a[j]; biasr[ni][4 + j] = c[j]; }
    } else if (PART && d.bias && nc < d.N) {                        // N % 8 != 0 (2- / 3-class heads): the channels past N store zeros
#pragma unroll
      for (int j = 0; j < 8; ++j) if (nc + j < d.N) biasr[ni][j] = d.bias[nn + j];
    }
  }
  const int G = gridDim.x, NS = ka.ns, LA = NS - 1;               // lookahead: tiles in flight beyond the current one
  const int ndma = ka.nslot * NCH;                                  // DMA instructions per wave per tile
  __syncthreads();                                                  // affine table visible; nothing in flight yet
  int t = blockIdx.x;
  for (int k = 0; k < LA; ++k)
    if (t + k * G < ka.tiles) issue(t + k * G, k);
  for (int it = 0; t < ka.tiles; t += G, ++it) {
    const int cur = it % NS;
    // tile t has landed once at most the DMA instructions of the tiles issued after it are outstanding
    int later = 0;
    for (int k = 1; k < LA; ++k) if (t + k * G < ka.tiles) ++later;
    vm_wait(later * ndma);
    lds_barrier();                       // ... for every wave; the stage of tile t - G is free (its MFMAs are done)
    if (t + LA * G < ka.tiles) issue(t + LA * G, (it + LA) % NS);
    if constexpr (AFF) {
      if (!(ka.dbg & 1)) transform(t, cur);
      if (!(ka.dbg & 2)) lds_barrier();
    }
    int b, oy0, ox0;
    tile_origin(t, b, oy0, ox0);
    const unsigned char* lds_h = smem + cur * stage;
#pragma unroll
    for (int ni = 0; ni < NTI; ++ni) {
      f32x4 acc[4][2];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) { acc[mf][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[mf][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      // flat (chunk, tap) sequence; the pixel fragments of step i+1 are requested before the MFMAs of step i, and a scheduling
      // barrier per step keeps the compiler from hoisting more of them (the weights already take 72-144 VGPRs)
      auto load_fa = [&](int step, u32x4 (&fa)[4]) {
        const int ch = step / TAPS, tp = step - ch * TAPS;
        const int toff = (tp / KW) * HW + (tp % KW);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int ap = a_base[mf] + toff;
          fa[mf] = *(const u32x4*)(lds_h + ch * hpb + ap * 64 + ((g ^ swz(ap)) << 4));
        }
      };
      u32x4 fa0[4], fa1[4];
      load_fa(0, fa0);
#pragma unroll
      for (int step = 0; step < NCH * TAPS; ++step) {
        const int ch = step / TAPS, tp = step - ch * TAPS;
        if (step + 1 < NCH * TAPS) { if (step & 1) load_fa(step + 1, fa0); else load_fa(step + 1, fa1); }
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const u32x4& f = (step & 1) ? fa1[mf] : fa0[mf];
          mma16<T>(acc[mf][0], wreg[ni][ch][tp][0], f);           // D = W * X^T
          mma16<T>(acc[mf][1], wreg[ni][ch][tp][1], f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- epilogue of (tile, column tile): lane (g, l15) owns channels nc .. nc+7 of pixel l15 of each 16-pixel row group -----
      const int nc = (blockIdx.y * NTI + ni) * BN + wn * 32 + g * 8;
      const bool nv = nc < d.N;
      int dd = 0, nn = nc;
      if (d.ps_cout > 0) { dd = nc / d.ps_cout; nn = nc - dd * d.ps_cout; }
      // destination of this lane's 8-channel group: the virtual concat of an input gradient (several destinations, each a whole
      // number of groups; one destination in every SNUNet launch: the scan is a scalar compare per extra destination)
      int di = 0;
      if constexpr (PART) {
#pragma unroll
        for (int k = 1; k < KSMI_MAX_SRC; ++k) if (k < d.ndst && nc >= d.dst[k].n_begin) di = k;
      }
      T* const obase = PART ? (T*)d.dst[di].ptr + d.dst[di].c_off + (nn - d.dst[di].n_begin) : (T*)d.dst[0].ptr + d.dst[0].c_off + nn;
      const int dC = PART ? d.dst[di].C : d.dst[0].C;
      const T* const mbase = (const T*)d.mask_src + nc;
      float mm[MASK ? 8 : 1], mr[MASK ? 8 : 1], mg[MASK ? 8 : 1], mb[MASK ? 8 : 1];
      const float (&bias)[8] = biasr[ni];
      auto ld8 = [&](const float* qp, float* o, int at) {
        const f32x4 a = *(const f32x4*)(qp + at), c = *(const f32x4*)(qp + at + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = c[j]; }
      };
      if constexpr (MASK) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { mm[j] = 0.f; mr[j] = 0.f; mg[j] = 0.f; mb[j] = 0.f; }
        if (nv) { ld8(d.m_mean, mm, nc); ld8(d.m_rstd, mr, nc); ld8(d.m_scale, mg, nc); ld8(d.m_shift, mb, nc); }
      }
      uint32_t opix[4];
      bool ok[4];
      u32x4 mv[MASK ? 4 : 1];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int p = wm * 64 + mf * 16 + l15;
        const int ly = dTW.div(p), lx = p - ly * d.TW;
        const int oy = oy0 + ly, ox = ox0 + lx;
        ok[mf] = nv && p < P && oy < d.Hout && ox < d.Wout;
        opix[mf] = d.ps_cout > 0 ? (uint32_t)((b * 2 * d.Hout + 2 * oy + (dd >> 1)) * (2 * d.Wout) + 2 * ox + (dd & 1))
                                 : (uint32_t)((b * d.Hout + oy) * d.Wout + ox);
        if constexpr (MASK) {
          mv[mf] = (u32x4){0u, 0u, 0u, 0u};
          if (ok[mf]) mv[mf] = *(const u32x4*)(mbase + (size_t)opix[mf] * d.N);
        }
      }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[mf][0][r] + bias[r]; v[4 + r] = acc[mf][1][r] + bias[4 + r]; }
        if constexpr (MASK) {
          float m[8];
          vec_unpack<T>(mv[MASK ? mf : 0], m);
#pragma unroll
          for (int j = 0; j < 8; ++j) if (!(m[j] * mg[j] + mb[j] > 0.f)) v[j] = 0.f;
        }
        // statistics of the value the later passes READ (the stored bf16), as BatchNorm in the reference sees the stored tensor:
        // sums of the fp32 accumulators describe a tensor nobody normalises (round 5; the gate epilogue of igemm4 always did)
        const u32x4 pk = vec_pack<T>(v);
        float q[8];
        vec_unpack<T>(pk, q);
        if constexpr (MASK) {
          float m[8];
          vec_unpack<T>(mv[MASK ? mf : 0], m);
#pragma unroll
          for (int j = 0; j < 8; ++j) if (ok[mf]) { ssum[j] += q[j]; ssq[j] += q[j] * ((m[j] - mm[j]) * mr[j]); }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (ok[mf]) { ssum[j] += q[j]; ssq[j] += q[j] * q[j]; }
        }
        if (ok[mf] && !(ka.dbg & 4)) *(u32x4*)(obase + (size_t)opix[mf] * dC) = pk;
      }
    }
  }
  // ---- statistics: one row per workgroup -----------------------------------------------------------------------------------
  if (NTI == 1 && d.stats) {
    __syncthreads();
    float* red = (float*)smem;                                      // [WN groups][4 waves][2][32]
    const int ws = wn * 4 + wm;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = row16_sum(ssum[j]), q = row16_sum(ssq[j]);
      if (l15 == 0) { red[(ws * 2 + 0) * 32 + g * 8 + j] = a; red[(ws * 2 + 1) * 32 + g * 8 + j] = q; }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, nn = tid - which * BN;
      const int grp = nn >> 5, n = nn & 31;
      const float* rg = red + (size_t)grp * 4 * 2 * 32;
      const float v = rg[(0 * 2 + which) * 32 + n] + rg[(1 * 2 + which) * 32 + n] + rg[(2 * 2 + which) * 32 + n] + rg[(3 * 2 + which) * 32 + n];
      const int ncol = blockIdx.y * BN + nn;
      if (ncol < d.Npad) d.stats[((size_t)blockIdx.x * 2 + which) * d.Npad + ncol] = v;
    }
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------------
bool ksmi_igemm3_geom(const ksmi_conv_desc* d, int dtype, ksmi_igemm3_geom_t* g) {
  static const bool off = ksmi_knob_is_set("KSMI_IGEMM3_OFF");
  if (off || dtype != KSMI_BF16) return false;
  const int taps = d->KH * d->KW;
  if (!((d->KH == 3 && d->KW == 3) || (d->KH == 1 && d->KW == 1) || (d->KH == 2 && d->KW == 2))) return false;
  if (d->uniform_kc || d->nchunks < 1 || d->nchunks > 8) return false;
  if (d->ndst < 1 || d->ndst > KSMI_MAX_SRC || d->dst[0].accumulate || d->dst[0].n_begin != 0) return false;
  // several destinations (input gradient of a virtual concat: FC-Siam conv12d -> 3 x 16 channels): whole 8-channel groups each, in order
  if (d->ndst > 1) {
    if (d->mask_src || d->ps_cout || (d->N % 8)) return false;
    int nb = 0;
    for (int k = 0; k < d->ndst; ++k) {
      const ksmi_dst& q = d->dst[k];
      if (q.accumulate || q.n_begin != nb || q.n_len < 8 || (q.n_len % 8) || (q.C % 8) || (q.c_off % 8) || q.c_off + q.n_len > q.C || ((uintptr_t)q.ptr & 15)) return false;
      nb += q.n_len;
    }
    if (nb != d->N) return false;
  }
  if (d->alpha != 0.f || d->resid || d->relu_out || d->out_sy || d->in_sy) return false;
  // KSMI_IGEMM3_PARTIAL=0: the round-4 rule (whole 32-channel chunks, N % 8 == 0, >= 32 padded columns) for same-box A/B runs
  static const bool partial_on = (ksmi_knob_int("KSMI_IGEMM3_PARTIAL", 1) != 0);
  if (!partial_on) {
    if ((d->N % 8) || d->Npad < 32 || d->ndst != 1) return false;
    for (int i = 0; i < d->nsrc; ++i) if (d->src[i].c_len % 32) return false;
  }
  {   // what needs the PART instances (3 x 3, and 2 x 2 with <= 2 chunks)
    bool part = d->ndst > 1 || (d->N % 8) != 0;
    for (int i = 0; i < d->nsrc; ++i) part = part || (d->src[i].c_len % 32) != 0;
    if (part && !(taps == 9 || (taps == 4 && d->nchunks <= 2))) return false;
  }
  if ((d->dst[0].C % 8) || (d->dst[0].c_off % 8) || d->Npad < 16) return false;
  // N % 8 != 0 (the 2- / 3-class heads, destination with a channel stride of 8): the last 8-channel group is stored whole, its pad
  // channels as zeros (zero weights, no bias) -- plain epilogue only, and the destination row has to hold the whole group
  if ((d->N % 8) && (d->mask_src || d->ps_cout || d->dst[0].c_off + ((d->N + 7) & ~7) > d->dst[0].C)) return false;
  if (d->ps_cout && (d->ps_cout % 8)) return false;
  // ReLU-mask + BN-backward sums epilogue: instances exist for 3x3 (KSMI_IGEMM3_MASK=1) but measured slower than the tile kernel /
  // igemm4 (K = 32: 154 vs 89 us, the mask loads spill 45 VGPRs next to the register-resident weights; K = 64: 72 vs 64 us): off
  static const bool mask_on = (ksmi_knob_int("KSMI_IGEMM3_MASK", 0) != 0);
  // ... except where the tile kernel cannot run at all (a source that is not whole 32-channel chunks: Unet decoder block 5, 16 channels
  // at 224 x 224 -- 424 us on the first-generation kernel)
  bool part_src = false;
  for (int i = 0; i < d->nsrc; ++i) part_src = part_src || (d->src[i].c_len % 32) != 0;
  // (KSMI_IGEMM3_MASK_PART=0: leave those to the tile kernel's uniform-partial-chunk route, igemm2.hip klen)
  static const bool mask_part_on = (ksmi_knob_int("KSMI_IGEMM3_MASK_PART", 1) != 0);
  if (d->mask_src && !((mask_on || (part_src && partial_on && mask_part_on)) && taps == 9)) return false;
  if (d->gate_src || (d->mask_src && d->src[0].scale)) return false;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!al16(d->dst[0].ptr) || !al16(d->bias)) return false;
  for (int i = 0; i < d->nsrc; ++i) {
    // whole 16-byte granules of the source row: c_len itself need not be a multiple of 32 (last chunk: zero-page granules) nor of 8
    // (the row's pad channels are read and meet zero weights: they must exist inside the row)
    if (d->src[i].c_len < 1 || (d->src[i].C % 8) || (d->src[i].c_off % 8) || d->src[i].c_off + ((d->src[i].c_len + 7) & ~7) > d->src[i].C) return false;
    if (!al16(d->src[i].ptr)) return false;
    if (d->src[i].scale && (i > 0 || d->nsrc != 1)) return false;
    if ((size_t)d->B * d->Hin * d->Win * (size_t)d->src[i].C * 2 >= ((size_t)1 << 32)) return false;
  }
  if ((size_t)d->B * d->Hout * d->Wout * (d->ps_cout > 0 ? 4 : 1) >= ((size_t)1 << 31)) return false;
  const int HH = (d->TH - 1) * d->stride + d->KH, HW = (d->TW - 1) * d->stride + d->KW;
  const int HP = HH * HW;
  if (d->TH * d->TW > 256 || HP > 512) return false;
  // instances (the ones that compile without scratch): 3x3 with 1 chunk (one or two channel groups) or 2 chunks (one channel group,
  // one wave per SIMD, weights partly in AGPRs); 2x2 with 1, 2 (4: one group) chunks; 1x1 with 2, 4, 8 chunks and up to 4 column
  // tiles walked inside a workgroup (weights of all of them in registers: x is then read once for every column tile)
  const bool aff = d->src[0].scale != nullptr;
  g->WN = d->Npad >= 64 ? 2 : 1;
  g->NTI = 1;
  if (taps == 9) {
    if (d->nchunks > 2) return false;
    // (fused-operand K = 64 on this kernel, KSMI_IGEMM3_AFF2=1: 99 us against 68 us on igemm4, one 4-wave workgroup per CU)
    static const bool aff2_on = (ksmi_knob_int("KSMI_IGEMM3_AFF2", 0) != 0);
    if (d->nchunks == 2) { if (aff && !aff2_on) return false; g->WN = 1; }
  } else if (taps == 4) {
    if (aff || (d->nchunks != 1 && d->nchunks != 2 && d->nchunks != 4)) return false;
    if (d->nchunks == 4) g->WN = 1;
  } else {
    if (aff || (d->nchunks != 2 && d->nchunks != 4 && d->nchunks != 8)) return false;
  }
  const int nthr = 256 * g->WN;
  const int bn = 32 * g->WN;
  const int ntiles = (d->Npad + bn - 1) / bn;
  if (taps == 1 && d->stats == nullptr) {
    if (d->nchunks == 2 && ntiles % 4 == 0) g->NTI = 4;
    else if (d->nchunks <= 4 && ntiles % 2 == 0) g->NTI = 2;
  }
  g->hpb = (HP * 64 + nthr * 16 - 1) / (nthr * 16) * (nthr * 16);
  g->nslot = g->hpb / (nthr * 16);
  g->stage = d->nchunks * g->hpb;
  const size_t LDS_CU = 160 * 1024;
  const size_t l2 = 2 * (size_t)g->stage + 1024, l3 = 3 * (size_t)g->stage + 1024;
  if (l2 > LDS_CU) return false;
  const int cap = (g->WN == 1 && d->nchunks * taps * g->NTI <= 9) ? 2 : 1;      // workgroups per CU by registers (= the kernel's __launch_bounds__)
  // three stages (two tiles in flight) when that does not cost a workgroup per CU
  if (cap == 2) g->ns = (2 * l3 <= LDS_CU) ? 3 : ((2 * l2 <= LDS_CU) ? 2 : (l3 <= LDS_CU ? 3 : 2));
  else g->ns = l3 <= LDS_CU ? 3 : 2;
  g->lds = g->ns == 3 ? l3 : l2;
  if (g->lds < (size_t)4 * 2 * 2 * 32 * g->WN * sizeof(float)) g->lds = (size_t)4 * 2 * 2 * 32 * g->WN * sizeof(float);
  const int tilesX = (d->Wout + d->TW - 1) / d->TW, tilesY = (d->Hout + d->TH - 1) / d->TH;
  g->tiles = d->B * tilesX * tilesY;
  int per_cu = (int)(LDS_CU / g->lds);
  if (per_cu > cap) per_cu = cap;
  const int cus = ksmi_knob_int("KSMI_IGEMM3_CUS", 256);      // (a knob: the tests shrink the grid to force many rounds)
  const int gy = ntiles / g->NTI;
  int gx = cus * per_cu / gy;
  if (gx < 1) gx = 1;
  if (gx > g->tiles) gx = g->tiles;
  // equalise: every workgroup walks ceil(tiles / gx) tiles -> the smallest grid with the same number of rounds
  const int rounds = (g->tiles + gx - 1) / gx;
  gx = (g->tiles + rounds - 1) / rounds;
  g->gx = gx; g->gy = gy;
  return true;
}

int ksmi_igemm3_launch(const ksmi_conv_desc* d, const ksmi_igemm3_geom_t* g, hipStream_t st) {
  Ig3Args ka;
  ka.d = *d;
  const int HW = (d->TW - 1) * d->stride + d->KW;
  const int tilesX = (d->Wout + d->TW - 1) / d->TW, tilesY = (d->Hout + d->TH - 1) / d->TH;
  ka.m_tw = fastdiv_magic(d->TW); ka.m_hw = fastdiv_magic(HW); ka.m_tx = fastdiv_magic(tilesX); ka.m_ty = fastdiv_magic(tilesY);
  ka.dbg = ksmi_knob_int("KSMI_IG3_DBG", 0);
  ka.tiles = g->tiles; ka.hpb = g->hpb; ka.nslot = g->nslot; ka.stage = g->stage; ka.ns = g->ns;
  const unsigned char* const zero_page = ig3_zero();
  if (!zero_page) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm3: zero page");
  ka.zero = zero_page;
  const dim3 grid(g->gx, g->gy);
  const bool aff = d->src[0].scale != nullptr;
  const int taps = d->KH * d->KW;
#define KSMI_G3M(KH_, KW_, NCH_, WN_, AFF_, NTI_, MASK_)                                             \
  do {                                                                                               \
    auto kfn = igemm3_kernel<KH_, KW_, NCH_, WN_, AFF_, NTI_, MASK_>; KSMI_NOTE(kfn);                                \
    if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256 * WN_), g->lds, st, ka);                                  \
    return ksmi_check_launch("igemm3");                                                              \
  } while (0)
#define KSMI_G3(KH_, KW_, NCH_, WN_, AFF_, NTI_)                                                     \
  do {                                                                                               \
    auto kfn = igemm3_kernel<KH_, KW_, NCH_, WN_, AFF_, NTI_, false>; KSMI_NOTE(kfn);                                \
    if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256 * WN_), g->lds, st, ka);                                  \
    return ksmi_check_launch("igemm3");                                                              \
  } while (0)
#define KSMI_G3W(KH_, KW_, NCH_, AFF_, NTI_)                                                         \
  do { if (g->WN == 2) KSMI_G3(KH_, KW_, NCH_, 2, AFF_, NTI_); else KSMI_G3(KH_, KW_, NCH_, 1, AFF_, NTI_); } while (0)
  // partial chunks / thin heads / several destinations: their own instances (3 x 3 and 2 x 2 only: ksmi_igemm3_geom)
  bool part = d->ndst > 1 || (d->N % 8) != 0;
  for (int i = 0; i < d->nsrc; ++i) part = part || (d->src[i].c_len % 32) != 0;
#define KSMI_G3P(KH_, KW_, NCH_, WN_, AFF_, MASK_)                                                   \
  do {                                                                                               \
    auto kfn = igemm3_kernel<KH_, KW_, NCH_, WN_, AFF_, 1, MASK_, 0, true>; KSMI_NOTE(kfn);                          \
    if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256 * WN_), g->lds, st, ka);                                  \
    return ksmi_check_launch("igemm3");                                                              \
  } while (0)
  if (part) {
    if (taps == 9 && d->mask_src) {
      if (aff) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm3: fused operand and mask epilogue together");
      if (d->nchunks == 1) { if (g->WN == 2) KSMI_G3P(3, 3, 1, 2, false, true); else KSMI_G3P(3, 3, 1, 1, false, true); }
      if (d->nchunks == 2) KSMI_G3P(3, 3, 2, 1, false, true);
    } else if (taps == 9) {
      if (d->nchunks == 1) {
        if (aff) { if (g->WN == 2) KSMI_G3P(3, 3, 1, 2, true, false); else KSMI_G3P(3, 3, 1, 1, true, false); }
        else { if (g->WN == 2) KSMI_G3P(3, 3, 1, 2, false, false); else KSMI_G3P(3, 3, 1, 1, false, false); }
      }
      if (d->nchunks == 2) { if (aff) KSMI_G3P(3, 3, 2, 1, true, false); else KSMI_G3P(3, 3, 2, 1, false, false); }
    } else if (taps == 4 && !aff) {
      if (d->nchunks == 1) { if (g->WN == 2) KSMI_G3P(2, 2, 1, 2, false, false); else KSMI_G3P(2, 2, 1, 1, false, false); }
      if (d->nchunks == 2) { if (g->WN == 2) KSMI_G3P(2, 2, 2, 2, false, false); else KSMI_G3P(2, 2, 2, 1, false, false); }
    }
    return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm3: no partial-chunk instance (geometry / launch mismatch)");
  }
#undef KSMI_G3P
  if (taps == 9 && d->mask_src) {
    if (aff) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm3: fused operand and mask epilogue together");
    if (d->nchunks == 1) { if (g->WN == 2) KSMI_G3M(3, 3, 1, 2, false, 1, true); else KSMI_G3M(3, 3, 1, 1, false, 1, true); }
    if (d->nchunks == 2) KSMI_G3M(3, 3, 2, 1, false, 1, true);
  }
#define KSMI_G3D(NCH_, WN_)                                                                          \
  do {                                                                                               \
    auto kfn = igemm3_kernel<3, 3, NCH_, WN_, false, 1, false, 1>; KSMI_NOTE(kfn);                                   \
    if (g->lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256 * WN_), g->lds, st, ka);                                  \
    return ksmi_check_launch("igemm3");                                                              \
  } while (0)
  if (taps == 9 && d->dir == 1 && !aff && !d->mask_src) {
    if (d->nchunks == 1) { if (g->WN == 2) KSMI_G3D(1, 2); else KSMI_G3D(1, 1); }
    if (d->nchunks == 2) KSMI_G3D(2, 1);
  }
#undef KSMI_G3D
  if (taps == 9) {
    if (d->nchunks == 1) { if (aff) KSMI_G3W(3, 3, 1, true, 1); else KSMI_G3W(3, 3, 1, false, 1); }
    if (d->nchunks == 2) { if (aff) KSMI_G3(3, 3, 2, 1, true, 1); else KSMI_G3(3, 3, 2, 1, false, 1); }
  } else if (taps == 1) {
    if (d->nchunks == 2) { if (g->NTI == 4) KSMI_G3W(1, 1, 2, false, 4); else if (g->NTI == 2) KSMI_G3W(1, 1, 2, false, 2); else KSMI_G3W(1, 1, 2, false, 1); }
    if (d->nchunks == 4) { if (g->NTI == 2) KSMI_G3W(1, 1, 4, false, 2); else KSMI_G3W(1, 1, 4, false, 1); }
    if (d->nchunks == 8) KSMI_G3W(1, 1, 8, false, 1);
  } else if (taps == 4) {
    if (d->nchunks == 1) KSMI_G3W(2, 2, 1, false, 1);
    if (d->nchunks == 2) KSMI_G3W(2, 2, 2, false, 1);
    if (d->nchunks == 4) KSMI_G3(2, 2, 4, 1, false, 1);
  }
#undef KSMI_G3W
#undef KSMI_G3
#undef KSMI_G3M
  return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm3: no instance (geometry / launch mismatch)");
}
