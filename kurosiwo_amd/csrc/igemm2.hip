// Implicit-GEMM convolution, software-pipelined variant (the default when every concat source
// is a whole number of 64-byte k-chunks).
//
// Per k-chunk a workgroup needs a halo tile (HP pixels x 64 B) and a weight slab (taps x BN x 64 B).
// Both LDS images are double-buffered and filled by LDS-DMA (`global_load_lds_dwordx4`): lane i of a
// wave instruction lands at base + 16*i, so the image is lane-linear and the bank swizzle is applied
// on the per-lane SOURCE address (cdna_hip_programming.md rule 21).  Zero padding comes for free:
// out-of-image halo positions are never written (exec-masked lanes) and the buffers are zeroed once.
// One barrier per chunk; the DMA of chunk c+1 is in flight while the MFMAs of chunk c run.
// With a fused BN-apply+ReLU operand (AFF) the halo goes through registers instead (transform,
// ds_write); the weights still use DMA.
#include <stdlib.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"
#include "igemm_epilogue.h"

namespace {

struct Igemm2Args {            // kernel argument: the public descriptor + launcher-computed constants
  ksmi_conv_desc d;
  uint32_t m_tw, m_hw, m_tx, m_ty;   // fastdiv magics of TW, halo width, tilesX, tilesY
  int dbg;
  int stages;                        // LDS stages: 2 = double-buffered k-chunks, 1 = single
  int klen;                          // channels present in EVERY k-chunk (32 bf16 / 16 fp32 unless all sources are one equal partial chunk,
                                     // round 5): the 16-byte k-groups past it are never fetched and stay zero like the padding positions
};

// WN = 1: 4 waves, each 64 pixels x BNW = 16*NT channels.  WN = 2: 8 waves = two channel groups sharing ONE halo image
// (M = 256 pixels, N = 2*BNW): the L2 -> LDS traffic per output channel halves for N >= 64 (measured: the DMA path moves
// 2.3x the unique bytes and is the resource the K loop waits on).
// LEAN (single-chunk convolutions, K <= 32 bf16 channels: 13 level-0 launches per SNUNet step at ~200 us each): these
// workgroups are one long dependent chain (address setup -> DMA wait -> 72 MFMAs -> epilogue, ~24k cycles) and only latency
// matters, so the variant trades the per-lane address table and the tap double-buffer for registers: <= 168 VGPRs and 40 KB
// of LDS let 3 workgroups share a CU instead of 2.
// MP = 2 (WN = 2, no AFF): the workgroup computes TWO pixel tiles (blockIdx.x*2, +1) per weight slab: the L2 -> LDS traffic of a
// k-chunk drops from 2 x (halo + slab) to (2 halos + slab) -- these launches are bound by that path, not by the MFMA pipe (25 %
// busy) -- and every barrier covers twice the MFMAs.
template <typename T, int NT, int KH, int KW, bool AFF, bool EX, int WN, bool LEAN = false, int MP = 1>
__global__ __launch_bounds__(256 * WN, LEAN ? 3 : 1) void igemm2_fwd_kernel(const Igemm2Args ka) {
  static_assert(MP == 1 || (!AFF && !LEAN), "two tiles per workgroup: DMA halo path only");
  const ksmi_conv_desc& d = ka.d;
  const long long tm0 = __builtin_readcyclecounter();
  constexpr int NTHR = 256 * WN;
  constexpr int TAPS = KH * KW;
  constexpr int BNW = NT * 16;                              // channels per wave
  constexpr int BN = BNW * WN;                              // channels per workgroup
  constexpr int VEC = ElemTraits<T>::kVec;
  constexpr int KC = VEC * 4;
  constexpr int WVEC = TAPS * BN * 4;                       // 16-byte vectors of one weight slab
  constexpr int WITER = (WVEC + NTHR - 1) / NTHR;
  constexpr int MAXSLOT = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;                  // pixel group / channel group of this wave
  const int g = lane >> 4, l15 = lane & 15;

  const int tilesX = (d.Wout + d.TW - 1) / d.TW, tilesY = (d.Hout + d.TH - 1) / d.TH;
  const FastDiv dTX(tilesX, ka.m_tx), dTY(tilesY, ka.m_ty);
  const int ntile = d.B * tilesX * tilesY;
  int tb[MP], toy[MP], tox[MP];
  bool tvalid[MP];
#pragma unroll
  for (int mp = 0; mp < MP; ++mp) {
    int tile = blockIdx.x * MP + mp;
    tvalid[mp] = tile < ntile;
    if (!tvalid[mp]) tile = ntile - 1;                      // (odd tile count: the second tile is computed and dropped)
    const int q1 = dTX.div(tile);
    const int tx = tile - q1 * tilesX;
    tb[mp] = dTY.div(q1);
    const int ty = q1 - tb[mp] * tilesY;
    toy[mp] = ty * d.TH; tox[mp] = tx * d.TW;
  }
  const int b = tb[0], oy0 = toy[0], ox0 = tox[0];
  const int n0 = blockIdx.y * BN;
  const int S = d.stride;
  const int HH = (d.TH - 1) * S + KH, HW = (d.TW - 1) * S + KW;
  const int HP = HH * HW;
  const int P = d.TH * d.TW;
  const int HPB = (HP * 64 + 1023) & ~1023;                 // halo bytes, whole wave-instructions (1 KiB)
  const int BUFB = MP * HPB + TAPS * BN * 64;               // one stage: MP halo images + the weight slab
  const int nslot = (HPB + NTHR * 16 - 1) / (NTHR * 16);    // NTHR lanes x 16 B per slot-iteration
  const FastDiv dHW(HW, ka.m_hw), dTW(d.TW, ka.m_tw);

  // ---- per-thread halo slots: LDS position v = s*256 + tid <-> (pixel v>>2, slot v&3) -----------
  // (branch-free: selects only; the strided input view exists only in the EX variant)
  int slot_goff[MAXSLOT];                                   // pixel index in the image, or -1
  int slot_goff2[MP > 1 ? MAXSLOT : 1];                     // ... of the second tile
  int slot_qb[MAXSLOT];                                     // DMA path: byte offset of the k-group fetched into this LDS slot
  int slot_lds[AFF ? MAXSLOT : 1];                          // register path: where k-group myq of the pixel goes
  const int iy0 = oy0 * S - d.pad, ix0 = ox0 * S - d.pad_x;
#pragma unroll
  for (int s = 0; s < MAXSLOT; ++s) {
    const int v = tid + s * NTHR;
    const int pix = v >> 2, sl = v & 3;
    const int hy = dHW.div(pix), hx = pix - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = v < HP * 4 && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win && (sl ^ swz(pix)) * VEC < ka.klen;
    int gp;
    if constexpr (EX) gp = src_pixel(d, b, iy, ix);
    else gp = (b * d.Hin + iy) * d.Win + ix;
    slot_goff[s] = ok ? gp : -1;
    if constexpr (MP > 1) {
      const int iy2 = toy[1] * S - d.pad + hy, ix2 = tox[1] * S - d.pad_x + hx;
      const bool ok2 = v < HP * 4 && (unsigned)iy2 < (unsigned)d.Hin && (unsigned)ix2 < (unsigned)d.Win && (sl ^ swz(pix)) * VEC < ka.klen;
      int gp2;
      if constexpr (EX) gp2 = src_pixel(d, tb[1], iy2, ix2);
      else gp2 = (tb[1] * d.Hin + iy2) * d.Win + ix2;
      slot_goff2[s] = ok2 ? gp2 : -1;
    }
    slot_qb[s] = (sl ^ swz(pix)) << 4;
    if constexpr (AFF) slot_lds[s] = pix * 64 + ((sl ^ swz(pix)) << 4);   // (register path: thread owns k-group sl)
  }
  const int myq = tid & 3;
  // padding positions are never written by the DMA: zero them once (both stages; interior patches have none)
  const bool two_stage = ka.stages > 1;
#pragma unroll
  for (int s = 0; s < MAXSLOT; ++s) {
    if (tid + s * NTHR < HP * 4 && slot_goff[s] < 0) {
      const int off = AFF ? slot_lds[AFF ? s : 0] : (s * NTHR + tid) * 16;
      *(u32x4*)(smem + off) = (u32x4){0u, 0u, 0u, 0u};
      if (two_stage) *(u32x4*)(smem + BUFB + off) = (u32x4){0u, 0u, 0u, 0u};
    }
    if constexpr (MP > 1) {
      if (tid + s * NTHR < HP * 4 && slot_goff2[s] < 0) {
        const int off = HPB + (s * NTHR + tid) * 16;
        *(u32x4*)(smem + off) = (u32x4){0u, 0u, 0u, 0u};
        if (two_stage) *(u32x4*)(smem + BUFB + off) = (u32x4){0u, 0u, 0u, 0u};
      }
    }
  }

  // ---- per-lane fragment addresses ---------------------------------------------------------------
  int a_addr[LEAN ? 1 : 4][LEAN ? 1 : TAPS];
  int a_base[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    int p = wm * 64 + mf * 16 + l15;
    if (p >= P) p = 0;
    const int ly = dTW.div(p), lx = p - ly * d.TW;
    const int base = ly * S * HW + lx * S;
    a_base[mf] = base;
    if constexpr (!LEAN) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ap = base + (t / KW) * HW + (t % KW);
        a_addr[mf][t] = ap * 64 + ((g ^ swz(ap)) << 4);
      }
    }
  }
  int b_addr[NT];
#pragma unroll
  for (int nf = 0; nf < NT; ++nf) {
    const int n = wn * BNW + nf * 16 + l15;
    b_addr[nf] = n * 64 + ((g ^ swz(n)) << 4);
  }
  // weight DMA: vector v = i*256 + tid -> row (tap*BN + n), slot; source k-group = slot ^ swz(n)
  int w_src[WITER];                                         // element offset inside one chunk's slab, or -1
#pragma unroll
  for (int i = 0; i < WITER; ++i) {
    const int v = i * NTHR + tid;
    const int row = v >> 2, sl = v & 3;
    const int t = row / BN, n = row - t * BN;
    // LDS row j = nf*16 + g*4 + r of a wave's A operand holds output channel g*4*NT + nf*4 + r of its channel group
    // (epilogue: igemm_epilogue.h, epi_col)
    const int nb = n / BNW, j = n - nb * BNW;
    const int nsrc_ = nb * BNW + ((j >> 2) & 3) * 4 * NT + (j >> 4) * 4 + (j & 3);
    w_src[i] = (v < WVEC && n0 + nsrc_ < d.Npad) ? ((t * d.Npad + n0 + nsrc_) * 64 + ((sl ^ swz(n)) << 4)) : -1;   // bytes
  }

  f32x4 acc[4][NT];
  f32x4 acc2[MP > 1 ? 4 : 1][NT];                           // second tile
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int nf = 0; nf < NT; ++nf) {
      acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (MP > 1) acc2[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  u32x4 hreg[AFF ? MAXSLOT : 1];

  // Per-chunk scalars of the source (base pointer of the chunk's first channel, bytes per pixel): fetched ONE chunk ahead of
  // their use so the scalar loads never stall the wave between the barrier and the MFMAs.  All per-lane offsets are 32-bit
  // (the launcher checks every source spans < 4 GiB) -> SGPR-base + VGPR-offset addressing, no 64-bit VALU in the loop.
  struct ChunkSrc { const unsigned char* sp; uint32_t cb; const float* scale; const float* shift; int relu; };
  auto chunk_scalars = [&](int ch) -> ChunkSrc {
    ChunkSrc c;
    const int si = chunk_src_of(d, ch);
    const ksmi_src& sr = d.src[si];
    c.sp = (const unsigned char*)((const T*)sr.ptr + sr.c_off + chunk_c0_of(d, ch));
    c.cb = (uint32_t)sr.C * (uint32_t)sizeof(T);
    if constexpr (AFF) { c.scale = sr.scale + chunk_c0_of(d, ch); c.shift = sr.shift + chunk_c0_of(d, ch); c.relu = sr.relu; }
    else { c.scale = nullptr; c.shift = nullptr; c.relu = 0; }
    return c;
  };
  const unsigned char* const wpk = (const unsigned char*)d.wpk;
  const uint32_t wslab = (uint32_t)TAPS * (uint32_t)d.Npad * 64u;   // bytes of one chunk's weight slab
  auto issue_weights = [&](int ch, int buf) {
    const unsigned char* wsrc = wpk + (size_t)ch * wslab;
    unsigned char* wdst = smem + buf * BUFB + MP * HPB;
#pragma unroll
    for (int i = 0; i < WITER; ++i) {
      if (w_src[i] >= 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (uint32_t)w_src[i]),
                                         (__attribute__((address_space(3))) void*)(wdst + (i * NTHR + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto issue_halo_dma = [&](const ChunkSrc& c, int buf) {
    unsigned char* hdst = smem + buf * BUFB;
#pragma unroll
    for (int s = 0; s < MAXSLOT; ++s) {
      const uint32_t off = (uint32_t)slot_goff[s] * c.cb + (uint32_t)slot_qb[s];
      if (s < nslot && slot_goff[s] >= 0)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(c.sp + off),
                                         (__attribute__((address_space(3))) void*)(hdst + (s * NTHR + wave * 64) * 16), 16, 0, 0);
    }
    if constexpr (MP > 1) {
#pragma unroll
      for (int s = 0; s < MAXSLOT; ++s) {
        const uint32_t off = (uint32_t)slot_goff2[s] * c.cb + (uint32_t)slot_qb[s];
        if (s < nslot && slot_goff2[s] >= 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(c.sp + off),
                                           (__attribute__((address_space(3))) void*)(hdst + HPB + (s * NTHR + wave * 64) * 16), 16, 0, 0);
      }
    }
  };
  auto load_halo_regs = [&](const ChunkSrc& c) {           // AFF: thread owns k-group myq of its slot pixels
#pragma unroll
    for (int s = 0; s < MAXSLOT; ++s) {
      const uint32_t off = (uint32_t)slot_goff[s] * c.cb + (uint32_t)(myq * 16);
      if (s < nslot && slot_goff[s] >= 0) hreg[AFF ? s : 0] = *(const u32x4*)(c.sp + off);
    }
  };
  auto store_halo_regs = [&](const ChunkSrc& c, int buf) {
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sc[j] = c.scale[myq * VEC + j]; sh[j] = c.shift[myq * VEC + j]; }
    unsigned char* hdst = smem + buf * BUFB;
#pragma unroll
    for (int s = 0; s < MAXSLOT; ++s)
      if (s < nslot && slot_goff[s] >= 0) {
        float f[VEC];
        vec_unpack<T>(hreg[AFF ? s : 0], f);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          f[j] = f[j] * sc[j] + sh[j];
          if (c.relu) f[j] = fmaxf(f[j], 0.f);
        }
        *(u32x4*)(hdst + slot_lds[AFF ? s : 0]) = vec_pack<T>(f);
      }
  };

  const long long tm1 = __builtin_readcyclecounter();
  __syncthreads();                                          // zero fill visible before any DMA lands
  // ---- prologue: stage chunk 0 ---------------------------------------------------------------------
  ChunkSrc cnext = chunk_scalars(0);
  issue_weights(0, 0);
  if constexpr (AFF) { load_halo_regs(cnext); store_halo_regs(cnext, 0); }
  else issue_halo_dma(cnext, 0);
  if (d.nchunks > 1) cnext = chunk_scalars(1);

  const int dbg = ka.dbg;            // profiling switches (KSMI_DBG): 1 no MFMA, 2 no DMA, 4 no epilogue
  for (int ch = 0; ch < d.nchunks; ++ch) {
    const int buf = two_stage ? (ch & 1) : 0;
    const int nbuf = two_stage ? (buf ^ 1) : 0;
    __syncthreads();                                        // chunk ch landed (vmcnt(0) before the barrier); the other stage is free
    const bool more = ch + 1 < d.nchunks;
    const ChunkSrc ccur = cnext;                            // scalars of chunk ch + 1
    if (two_stage && more && !(dbg & 2)) {
      issue_weights(ch + 1, nbuf);
      if constexpr (AFF) load_halo_regs(ccur);
      else issue_halo_dma(ccur, nbuf);
    }
    if (ch + 2 < d.nchunks) cnext = chunk_scalars(ch + 2);  // in flight during the MFMAs
    const unsigned char* lds_halo = smem + buf * BUFB;
    const unsigned char* lds_w = lds_halo + MP * HPB;
    if constexpr (LEAN) {
      if (!(dbg & 1)) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int toff = (t / KW) * HW + (t % KW);
          u32x4 fa1[4], fb1[NT];
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) {
            const int ap = a_base[mf] + toff;
            fa1[mf] = *(const u32x4*)(lds_halo + ap * 64 + ((g ^ swz(ap)) << 4));
          }
#pragma unroll
          for (int nf = 0; nf < NT; ++nf) fb1[nf] = *(const u32x4*)(lds_w + t * BN * 64 + b_addr[nf]);
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NT; ++nf) mma16<T>(acc[mf][nf], fb1[nf], fa1[mf]);
        }
      }
    } else if constexpr (MP > 1) {
      if (!(dbg & 1)) {
        // steps u = tap * 2 + tile: the pixel fragments of step u+1 (and, entering a new tap, its weight fragments) are requested
        // before the MFMAs of step u
        u32x4 fa[2][4], fb[2][NT];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) fa[0][mf] = *(const u32x4*)(lds_halo + a_addr[mf][0]);
#pragma unroll
        for (int nf = 0; nf < NT; ++nf) fb[0][nf] = *(const u32x4*)(lds_w + b_addr[nf]);
#pragma unroll
        for (int u = 0; u < TAPS * 2; ++u) {
          const int t = u >> 1, mp = u & 1, cur = u & 1, nxt = cur ^ 1, wc = t & 1;
          if (u + 1 < TAPS * 2) {
            const int t1 = (u + 1) >> 1, mp1 = (u + 1) & 1;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) fa[nxt][mf] = *(const u32x4*)(lds_halo + mp1 * HPB + a_addr[mf][t1]);
            if (mp1 == 0) {
#pragma unroll
              for (int nf = 0; nf < NT; ++nf) fb[wc ^ 1][nf] = *(const u32x4*)(lds_w + t1 * BN * 64 + b_addr[nf]);
            }
          }
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NT; ++nf) {
              if (mp == 0) mma16<T>(acc[mf][nf], fb[wc][nf], fa[cur][mf]);
              else mma16<T>(acc2[mf][nf], fb[wc][nf], fa[cur][mf]);
            }
        }
      }
    } else if (!(dbg & 1)) {
    // fragments of tap t+1 are fetched from LDS while the MFMAs of tap t issue (one wave per SIMD:
    // nothing else hides the LDS latency)
    u32x4 fa[2][4], fb[2][NT];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) fa[0][mf] = *(const u32x4*)(lds_halo + a_addr[mf][0]);
#pragma unroll
    for (int nf = 0; nf < NT; ++nf) fb[0][nf] = *(const u32x4*)(lds_w + b_addr[nf]);
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      constexpr int dummy = 0; (void)dummy;
      const int cur = t & 1, nxt = cur ^ 1;
      if (t + 1 < TAPS) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) fa[nxt][mf] = *(const u32x4*)(lds_halo + a_addr[mf][t + 1]);
#pragma unroll
        for (int nf = 0; nf < NT; ++nf) fb[nxt][nf] = *(const u32x4*)(lds_w + (t + 1) * BN * 64 + b_addr[nf]);
      }
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NT; ++nf) mma16<T>(acc[mf][nf], fb[cur][nf], fa[cur][mf]);   // D = W * X^T (see epilogue)
    }
    }
    if (two_stage) {
      if constexpr (AFF) { if (more) store_halo_regs(ccur, nbuf); }
    } else if (more) {
      // single stage (40 KB: more workgroups per CU instead of intra-workgroup overlap): refill after the MFMAs
      __syncthreads();
      issue_weights(ch + 1, 0);
      if constexpr (AFF) { load_halo_regs(ccur); store_halo_regs(ccur, 0); }
      else issue_halo_dma(ccur, 0);
    }
  }
  const long long tm2 = __builtin_readcyclecounter();
  if (!(dbg & 4)) {
    // EX == false on the bf16 NT = 2 tile means the launcher proved the preconditions of the lean epilogue
    if constexpr (MP > 1) {
      // (tile validity is workgroup-uniform: the barriers inside the statistics epilogue stay convergent)
      if constexpr (!EX && NT == 2 && sizeof(T) == 2) {
        igemm_epilogue_fast<WN>(d, acc, smem, tid, wm, g, l15, tb[0], toy[0], tox[0], n0 + wn * BNW, P, ka.m_tw, wn, n0, blockIdx.x * 2);
        if (tvalid[1]) igemm_epilogue_fast<WN>(d, acc2, smem, tid, wm, g, l15, tb[1], toy[1], tox[1], n0 + wn * BNW, P, ka.m_tw, wn, n0, blockIdx.x * 2 + 1);
      } else {
        igemm_epilogue_direct<T, NT, EX, WN>(d, acc, smem, tid, wm, g, l15, tb[0], toy[0], tox[0], n0 + wn * BNW, P, ka.m_tw, wn, n0, blockIdx.x * 2);
        if (tvalid[1])
          igemm_epilogue_direct<T, NT, EX, WN>(d, acc2, smem, tid, wm, g, l15, tb[1], toy[1], tox[1], n0 + wn * BNW, P, ka.m_tw, wn, n0, blockIdx.x * 2 + 1);
      }
    } else if constexpr (!EX && NT == 2 && sizeof(T) == 2) igemm_epilogue_fast<WN>(d, acc, smem, tid, wm, g, l15, b, oy0, ox0, n0 + wn * BNW, P, ka.m_tw, wn, n0);
    else igemm_epilogue_direct<T, NT, EX, WN>(d, acc, smem, tid, wm, g, l15, b, oy0, ox0, n0 + wn * BNW, P, ka.m_tw, wn, n0);
  }
  if ((dbg & 8) && d.stats && tid == 0 && blockIdx.y == 0) {   // per-block phase timestamps (profiling only; clobbers stats)
    const long long tm3a = __builtin_readcyclecounter();      // epilogue instructions issued
    __builtin_amdgcn_s_waitcnt(0);
    const long long tm3 = __builtin_readcyclecounter();       // ... and its stores acknowledged
    long long* o = (long long*)d.stats + (size_t)blockIdx.x * 8;
    o[0] = tm0; o[1] = tm1; o[2] = tm2; o[3] = tm3; o[4] = tm3a;
  }
}

template <typename T>
int launch2(const ksmi_conv_desc* d, int dbg, hipStream_t st) {
  const int taps = d->KH * d->KW;
  const int tilesX = (d->Wout + d->TW - 1) / d->TW, tilesY = (d->Hout + d->TH - 1) / d->TH;
  const int gm = d->B * tilesX * tilesY;
  const int HH = (d->TH - 1) * d->stride + d->KH, HW = (d->TW - 1) * d->stride + d->KW;
  const int HP = HH * HW;
  if (d->TH * d->TW > 256 || HP * 4 > 2048) return ksmi_fail(KSMI_E_ARG, "conv: patch too large (TH*TW<=256, halo<=512 px)");
  for (int i = 0; i < d->nsrc; ++i) {   // per-lane source offsets are 32-bit
    const size_t px = d->in_sy ? (size_t)d->B * d->in_H * d->in_W : (size_t)d->B * d->Hin * d->Win;
    if (px * (size_t)d->src[i].C * sizeof(T) >= ((size_t)1 << 32)) return ksmi_fail(KSMI_E_UNSUPPORTED, "conv: a source tensor of 4 GiB or more is not supported");
  }
  static const int nt_cap = ksmi_knob_int("KSMI_NT_CAP", 2);      // BN=32: 80 KB of LDS = 2 workgroups per CU beats the BN=64 tile (1 per CU) by 15-35 %
  int nt = d->Npad >= 64 ? 4 : (d->Npad >= 32 ? 2 : 1);
  // token GEMMs (1x1) keep both a small halo (16 KB) and a small weight slab: BN = 64 still leaves 2 workgroups per CU
  static const int nt_cap11 = ksmi_knob_int("KSMI_NT_CAP_1X1", 2);
  if (nt > (taps == 1 ? nt_cap11 : nt_cap)) nt = taps == 1 ? nt_cap11 : nt_cap;
  // two channel groups per workgroup (8 waves, one halo image): 3x3 / 2x2 / 1x1 with at least 64 output channels
  static const bool wn_off = ksmi_knob_is_set("KSMI_WN1");
  static const bool lean_off = ksmi_knob_is_set("KSMI_NO_LEAN");
  const bool lean = !lean_off && d->nchunks == 1 && taps == 9 && nt == 2 && HP * 64 <= 24576 && d->src[0].scale == nullptr;   // (the AFF path spills at 168 VGPRs)
  static const int wn_min = ksmi_knob_int("KSMI_WN_MIN", 64);
  const int wn = (!lean && !wn_off && nt == 2 && d->Npad >= wn_min && taps <= 9) ? 2 : 1;
  const int bn = nt * 16 * wn;
  const int gy = (d->Npad + bn - 1) / bn;
  const size_t hpb = ((size_t)HP * 64 + 1023) & ~(size_t)1023;
  // single-chunk convolutions (K <= 32 bf16 channels) need one stage only: 40 KB -> 3-4 workgroups per CU
  static const int stage_cap = ksmi_knob_int("KSMI_STAGES", 2);
  const int stages = (d->nchunks > 1 && stage_cap > 1) ? 2 : 1;
  const bool aff = d->src[0].scale != nullptr;
  // two pixel tiles per workgroup and weight slab (MP = 2): long-K 3x3 convolutions on the 8-wave tile with >= 256 output channels
  // (ChangeFormer's 256-channel 224^2 / 112^2 layers: +1.8 % on its step; neutral to -0.5 % on SNUNet's 64..128-channel layers, which
  // keep one tile), when the halved grid still gives every CU >= 1.5 workgroups and two stages of (2 halos + slab) fit 160 KB of LDS
  static const bool mp_off = ksmi_knob_is_set("KSMI_IGEMM2_MP1");
  static const int mp_min_k = ksmi_knob_int("KSMI_IGEMM2_MP_MINK", 4);
  const bool mp2 = !mp_off && sizeof(T) == 2 && wn == 2 && nt == 2 && !aff && stages == 2 && taps == 9 && d->nchunks >= mp_min_k &&
                   2 * (2 * hpb + (size_t)taps * bn * 64) <= (size_t)160 * 1024 && gy >= 4 && (size_t)((gm + 1) / 2) * gy >= 384;
  const dim3 grid(mp2 ? (gm + 1) / 2 : gm, gy);
  size_t lds = stages * ((mp2 ? 2 : 1) * hpb + (size_t)taps * bn * 64);
  if (lds < (size_t)4 * wn * 2 * (bn / wn) * sizeof(float)) lds = (size_t)4 * wn * 2 * (bn / wn) * sizeof(float);
  // epilogue extras (scale, residual, ReLU, strided placement) compile into a separate kernel: the common path stays lean
  bool extras = d->alpha != 0.f || d->resid != nullptr || d->relu_out != 0 || d->out_sy != 0 || d->in_sy != 0;
  if (sizeof(T) == 2 && nt == 2) {
    // the lean epilogue (igemm_epilogue.h: igemm_epilogue_fast) replaces the general one when its preconditions hold
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const ksmi_dst& d0 = d->dst[0];
    const bool fast = d->ndst == 1 && (d->ps_cout == 0 || ((d->ps_cout % 8) == 0 && d->mask_src == nullptr)) && (d->N % 8) == 0 && d0.n_begin == 0 && (d0.C % 8) == 0 && (d0.c_off % 8) == 0 &&
                      al16(d0.ptr) && al16(d->bias) && al16(d->mask_src) && al16(d->m_mean) && al16(d->m_rstd) && al16(d->m_scale) &&
                      al16(d->m_shift) && (size_t)d->B * d->Hout * d->Wout * (d->ps_cout > 0 ? 4 : 1) < ((size_t)1 << 31);
    if (!fast) extras = true;
  }
  Igemm2Args ka;
  ka.d = *d;
  ka.m_tw = fastdiv_magic(d->TW); ka.m_hw = fastdiv_magic(HW); ka.m_tx = fastdiv_magic(tilesX); ka.m_ty = fastdiv_magic(tilesY);
  ka.dbg = dbg;
  ka.stages = stages;
  {
    constexpr int VEC_ = ElemTraits<T>::kVec, KC_ = VEC_ * 4;
    ka.klen = d->src[0].c_len % KC_ ? ((d->src[0].c_len % KC_ + VEC_ - 1) / VEC_) * VEC_ : KC_;      // (ksmi_igemm2_eligible: the same in every chunk)
  }
#define KSMI_L2W(NT_, KH_, KW_, AFF_, EX_, WN_)                                                     \
  do {                                                                                              \
    auto kfn = igemm2_fwd_kernel<T, NT_, KH_, KW_, AFF_, EX_, WN_>; KSMI_NOTE(kfn);                                 \
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(256 * WN_), lds, st, ka);                                    \
  } while (0)
#define KSMI_L2MP(NT_, KH_, KW_, EX_)                                                               \
  do {                                                                                              \
    auto kfn = igemm2_fwd_kernel<T, NT_, KH_, KW_, false, EX_, 2, false, 2>; KSMI_NOTE(kfn);                        \
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, ka);                                          \
  } while (0)
#define KSMI_L2LEAN(NT_, KH_, KW_, AFF_, EX_)                                                       \
  do {                                                                                              \
    auto kfn = igemm2_fwd_kernel<T, NT_, KH_, KW_, AFF_, EX_, 1, true>; KSMI_NOTE(kfn);                             \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, ka);                                          \
  } while (0)
#define KSMI_L2X(NT_, KH_, KW_, AFF_, EX_)                                                          \
  do {                                                                                              \
    if constexpr (NT_ == 2 && KH_ * KW_ == 9) {                                                     \
      if (lean) KSMI_L2LEAN(NT_, KH_, KW_, AFF_, EX_);                                              \
      else if (wn == 2) {                                                                           \
        bool done_ = false;                                                                         \
        if constexpr (!AFF_ && sizeof(T) == 2) { if (mp2) { KSMI_L2MP(NT_, KH_, KW_, EX_); done_ = true; } } \
        if (!done_) KSMI_L2W(NT_, KH_, KW_, AFF_, EX_, 2);                                          \
      } else KSMI_L2W(NT_, KH_, KW_, AFF_, EX_, 1);                                                 \
    } else if constexpr (NT_ == 2 && KH_ * KW_ < 9) { if (wn == 2) KSMI_L2W(NT_, KH_, KW_, AFF_, EX_, 2); else KSMI_L2W(NT_, KH_, KW_, AFF_, EX_, 1); } \
    else KSMI_L2W(NT_, KH_, KW_, AFF_, EX_, 1);                                                     \
  } while (0)
#define KSMI_L2(NT_, KH_, KW_, AFF_)                                                                \
  do { if (extras) KSMI_L2X(NT_, KH_, KW_, AFF_, true); else KSMI_L2X(NT_, KH_, KW_, AFF_, false); } while (0)
#define KSMI_D2(KH_, KW_)                                                                           \
  switch (nt) {                                                                                     \
    case 1: if (aff) KSMI_L2(1, KH_, KW_, true); else KSMI_L2(1, KH_, KW_, false); break;           \
    case 2: if (aff) KSMI_L2(2, KH_, KW_, true); else KSMI_L2(2, KH_, KW_, false); break;           \
    default: if (aff) KSMI_L2(4, KH_, KW_, true); else KSMI_L2(4, KH_, KW_, false); break;          \
  }
  if (d->KH == 3 && d->KW == 3) { KSMI_D2(3, 3) }
  else if (d->KH == 1 && d->KW == 1) { KSMI_D2(1, 1) }
  else if (d->KH == 2 && d->KW == 2) { KSMI_D2(2, 2) }
  else if (d->KH == 4 && d->KW == 4) { KSMI_D2(4, 4) }
  else return ksmi_fail(KSMI_E_UNSUPPORTED, "conv: kernel size not supported");
#undef KSMI_D2
#undef KSMI_L2
#undef KSMI_L2X
#undef KSMI_L2W
#undef KSMI_L2LEAN
#undef KSMI_L2MP
  return ksmi_check_launch("igemm2_fwd");
}

}  // namespace

// eligibility: every source is a whole number of k-chunks and only source 0 may carry an affine
bool ksmi_igemm2_eligible(const ksmi_conv_desc* d, int dtype) {
  const int kc = dtype == KSMI_BF16 ? 32 : 16;
  // whole chunks -- or (round 5, bf16) every source ONE partial chunk of the same width (16 + 16 + 16 channels: FC-Siam conv12d; a
  // 16-channel input gradient with the mask epilogue): the k-groups past it are never fetched (Igemm2Args.klen); no fused operand then
  static const bool part_on = (ksmi_knob_int("KSMI_IGEMM2_PARTIAL", 1) != 0);
  bool whole = true, part = dtype == KSMI_BF16 && part_on;
  for (int i = 0; i < d->nsrc; ++i) {
    const ksmi_src& q = d->src[i];
    if (q.c_len % kc) whole = false;
    if (q.c_len >= kc || q.c_len != d->src[0].c_len || (q.C % 8) || (q.c_off % 8) || q.c_off + ((q.c_len + 7) & ~7) > q.C || q.scale) part = false;
    if (i > 0 && q.scale) return false;
  }
  if (!whole && !part) return false;
  if (d->src[0].scale && d->nsrc != 1) return false;
  return true;
}

int ksmi_igemm2_launch(const ksmi_conv_desc* d0, int dtype, hipStream_t st) {
  static const int dbg = ksmi_knob_int("KSMI_DBG", 0);
  if (dtype == KSMI_BF16) return launch2<bf16_t>(d0, dbg, st);
  return launch2<float>(d0, dbg, st);
}
