// Implicit-GEMM 3x3 (stride 1, pad 1) convolution for LONG K and >= 64 output channels, bf16: persistent workgroups, everything
// streamed through two LDS rings by asm LDS-DMA with counted vmcnt (dma.h), never drained inside the K loop.
//
// Why a third kernel (measured on SNUNet bs = 32, profiles/r03_*): the mid-level layers (K = N = 64 .. 512, conv2 of
// conv_block_nested and its input gradient; the dense-skip conv1 layers with K = 256 .. 1024) ran at 14 - 30 % of the MFMA peak on
// igemm2.hip.  There a workgroup is ONE dependent chain per tile -- tables, zero fill, DMA, `vmcnt(0)` + barrier per 32-channel
// k-chunk, epilogue -- and with one 8-wave workgroup per CU nothing overlaps the chain's latencies; the weight slab of a chunk
// (9 taps x BN x 64 B = 37 .. 74 KB) also made two stages the most LDS could hold.
// Here
//   * the K loop is a flat sequence of steps s = (chunk, kernel row r): 3 taps = 12 * NF MFMAs per wave and ONE raw s_barrier per
//     step; the weight ring holds 3 steps (slot = r, 3 taps x BN x 64 B), the halo ring 3 chunks (slot = chunk % 3, 24 KB);
//   * at the top of step s the weights of step s + 2 and (r == 0) the halo of chunk + 2 are issued; the wait in front of a step is
//     `vmcnt(nW + [r != 0] * nH)` = exactly the DMA instructions issued after the weights of that step (dma.h: reads return in
//     order), so two steps of weights and two chunks of halo are always in flight across the barriers;
//   * workgroups are persistent over pixel tiles (t = first + k * gx): the rings run straight through the tile boundary (the next
//     tile's first halo chunks and weight rows land during the epilogue), and BatchNorm statistics stay in registers across tiles
//     (one `stats` row per workgroup, ksmi_conv_stats_rows);
//   * every wave issues the same number of DMA instructions per step whatever the tile: padding positions and the tail of the
//     ring read a zero page, so nothing is exec-masked, zero-filled or counted per tile;
//   * 8 waves = 4 pixel groups (64 pixels) x 2 column groups (16 * NF columns): the halo image is read from HBM/L2 once per
//     128 (NF = 4) or 64 (NF = 2) output channels, and the column tiles of one pixel tile run on the same XCD.
// A fused BN-apply + ReLU operand (conv2 reads relu(bn1(i)), models/snunet.py:24-25) is transformed in LDS one step before its
// chunk is first read (no extra barrier: the step barrier that follows publishes it).  Epilogue = the lean epilogue of
// igemm_epilogue.h (bias, accumulate, ReLU-mask + BN-backward sums) restated for persistence, 16-byte stores.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"
#include "igemm_epilogue.h"
#include "igemm4.h"
#include "dma.h"

namespace {

template <int N> __device__ __forceinline__ void vm_wait_c() {         // counted wait with a compile-time count
  static_assert(N >= 0 && N < 64, "vmcnt field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __attribute__((aligned(64))) unsigned char ig4_zero_page[64];      // zero-initialised device memory
KSMI_DEVICE_SYMBOL_GETTER(ig4_zero, ig4_zero_page)

struct Ig4Args {
  ksmi_conv_desc d;
  uint32_t m_tw, m_hw, m_tx, m_ty;
  int th, tw;                      // output patch of a workgroup (th * tw <= 64 * WM pixels)
  int hslot, nh, nhs;              // halo ring: bytes per slot (8 KiB granules = one DMA piece per wave), pieces per wave, slots (2 | 3)
  int tiles, gx, gy;
  int rot;                         // rotated schedule: the step barrier sits in front of the previous step's last tap (see run_tiles_rot)
  int stagger;                     // waves 4-7 issue their DMA pieces after the first tap of a step instead of at its top
  int dbg;                         // KSMI_IG4_DBG profiling switches (wrong results): 1 no MFMA / fragment reads, 2 no DMA in the loop, 4 no
                                   // epilogue stores, 8 broadcast fragment reads (no LDS bandwidth), 16 no step barrier
  const unsigned char* zero;       // >= 16 zero bytes
};

constexpr int IG4_NHMAX = 5;       // halo pieces per wave at most: 40 KiB slot = 640 halo pixels

// WM pixel groups (64 pixels each) x WN = 8 / WM column groups (16 * NF columns each)
// 16-byte k-group slot swizzle of the halo image, keyed by the halo COLUMN: ds_read_b128 is served in the lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): 8 pixels of a fragment row read k-group g and the 8 pixels between
// them g ^ 1, and pixels 4 apart share the 64 banks of a 256-byte row.  h(x) = ((x >> 2) & 1) << 1 gives the four same-bank pixels
// of a group the slots {h, h^1^2.., } = four different ones for EVERY alignment of the 16 pixels (tap kx shifts them by 0..2); the
// `swz` of igemm_epilogue.h is conflict-free only for aligned rows (4 -> 6.7 LDS cycles per read on the 16 x 16 patch).
__device__ __forceinline__ int swz_h(int hx) { return ((hx >> 2) & 1) << 1; }

// EPI: 0 plain, 1 ReLU-mask + BatchNorm1-backward sums (mask_src), 2 gate: total gradient, ReLU gate of the block output and
// BatchNorm2-backward sums (gate_src / xhat_src; ksmi.h)
// DIR: 1 = input-gradient launch of the plain variant (ksmi_conv_desc.dir): a name tag for profilers, no code difference
// ROT: the rotated K-loop schedule (run_tiles_rot) instead of the per-role one (one schedule per instantiation: both in one kernel
// spill hundreds of registers)
// NWV: waves per workgroup.  8 = one 512-thread workgroup per CU (two waves per SIMD in lock step of the step barriers).  4 (round 5) =
// 256-thread workgroups, TWO per CU, each with its own rings (halo ring of 2 slots): the two workgroups of a CU drift apart, so the
// epilogue / store acknowledgement / tile prologue / barrier skew of one is covered by the K loop of the other.
// KH x KW: taps.  3 x 3 everywhere except the 2 x 2 stride-1 PHASE convolutions of ConvTranspose2d(k4, s2, p1) and of its input gradient
// (models/changeformer.py:329-336; plan_base._deconv / _deconv_bwd: strided output / input views, per-phase padding 0 | 1), which
// run on the chunk-granular schedule (ROT == 3) only.
template <int WM, int NF, bool AFF, int EPI, bool DBG = false, int DIR = 0, int ROT = 0, int NWV = 8, int KH = 3, int KW = 3>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void igemm4_kernel(const Ig4Args ka) {
  typedef bf16_t T;
  constexpr bool MASK = EPI == 1, GATE = EPI == 2;
  const ksmi_conv_desc& d = ka.d;
  constexpr int NT = 64 * NWV;                 // threads
  constexpr int WN = NWV / WM;
  constexpr int BNW = 16 * NF;                 // columns per wave
  constexpr int BN = WN * BNW;                 // columns per workgroup
  constexpr int NG = NF / 2;                   // 32-column groups per wave (epilogue ownership: 8 consecutive channels per lane)
  static_assert((KH == 3 && KW == 3) || (KH == 2 && KW == 2 && ROT == 3), "taps");
  constexpr int WSLOT = KW * BN * 64;          // one kernel row of taps
  constexpr int WPIECES = WSLOT / 1024;        // 24 (BN = 128) / 12 (BN = 64) / 6 (BN = 32)
  constexpr int WK = (WPIECES + NWV - 1) / NWV;   // pieces per wave (at most)
  constexpr int NHM = NWV == 4 ? 6 : (WM == 4 ? 3 : IG4_NHMAX);   // halo pieces per wave and slot: 24 KiB (<= 384 halo pixels) / 40 KiB (<= 640)
  constexpr int HSLOT = NHM * NT * 16;
  constexpr int nh = NHM;
  // CHUNK (ROT == 3, round 5): ONE wait + barrier per 32-channel chunk instead of one per kernel row.  The weights of a whole chunk
  // (3 rows) are resident when the chunk starts (weight ring = 2 chunks x 3 rows), so a wave issues 9 taps = 9 * 4 * NF MFMAs between
  // two barriers: at NF = 2 a row step is 24 MFMAs per wave (384 cycles) against ~520 cycles of wait + barrier + first-fragment latency
  // (cycle stamps, profiles/r05_ig4_phases.txt) -- the 32 / 64-column kernels were bound by the step skeleton, not by MFMAs or LDS.
  constexpr bool CHUNK = ROT == 3;
  constexpr int NHS = (NWV == 4 || (WM == 8 && NF == 4) || (CHUNK && WM == 8)) ? 2 : 3;   // halo ring slots (what 160 KiB -- 80 KiB for NWV = 4 -- holds next to the weight ring)
  // DEEP (round 5): weight ring of 5 step slots, weights issued FOUR steps ahead (run_tiles_deep).  vmcnt completes in issue order, so
  // the wait for the weights of a step also forces every older DMA: with the 3-slot ring (weights two steps ahead) a halo chunk had
  // ONE chunk time (3 steps) to land however many ring slots it owned, and the 32-column level-0 layers ran their K loop at the
  // memory latency (cycle stamps: K loop with DMA alone = full K loop = 5.3 k cycles per chunk; with MFMAs alone 4.1 k;
  // profiles/r05_ig4_phases.txt).  Four steps of weight lead give a halo chunk five steps.
  constexpr bool DEEP = ROT == 2;              // (launcher: only where !AFF, NHS == 3, NWV == 8, BN <= 64)
  static_assert(!DEEP || (!AFF && !DBG && NHS == 3 && NWV == 8 && BN <= 64), "deep weight ring: 32 / 64-column tiles with three halo slots");
  static_assert(!CHUNK || (!AFF && !DBG && NWV == 8 && (NF == 2 || KH == 2)), "chunk-granular schedule: the NF = 2 tiles and the 2 x 2 phase convolutions");
  constexpr int NWS = CHUNK ? 2 * KH : (DEEP ? 5 : 3);   // weight ring slots (one kernel row of taps each)
  constexpr int nhs = NHS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  long long stamp[8];                                                  // (DBG & 128: phase time stamps of workgroup 0, written over `stats`)
  int nstamp = 0;
  auto STAMP = [&]() { if (DBG && nstamp < 8) stamp[nstamp++] = __builtin_readcyclecounter(); };
  STAMP();
  const int g = lane >> 4, l15 = lane & 15;
  // workgroup -> (pixel-axis workgroup, column tile): the column tiles of a pixel workgroup sit on one XCD (block b runs on XCD b % 8)
  // and the pixel workgroups of an XCD walk neighbouring tiles (shared halo columns hit that XCD's L2).  Speed only.
  int pxw, nt;
  {
    const int id = blockIdx.x;
    if ((ka.gx & 7) == 0) { const int xcd = id & 7, q = id >> 3; nt = q % ka.gy; pxw = xcd * (ka.gx >> 3) + q / ka.gy; }
    else { nt = id % ka.gy; pxw = id / ka.gy; }
  }
  const int n0 = nt * BN;
  const int TH = ka.th, TW = ka.tw;
  const int HW = TW + KW - 1;
  const int HP = (TH + KH - 1) * HW;
  const int P = TH * TW;
  const int tilesX = (d.Wout + TW - 1) / TW, tilesY = (d.Hout + TH - 1) / TH;
  const FastDiv dTX(tilesX, ka.m_tx), dTY(tilesY, ka.m_ty), dHW(HW, ka.m_hw), dTW(TW, ka.m_tw);
  const int nch = d.nchunks;
  const int dbg = DBG ? ka.dbg : 0;                                    // (the switches exist only in the DBG instantiation)
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned char* const wring = smem + nhs * HSLOT;
  constexpr int WPAD = (WPIECES % NWV) ? 4096 : 0;                     // landing pad of the dummy weight pieces
  float* const st_tab = (float*)(wring + NWS * WSLOT + WPAD);                 // [waves][2][BNW] statistics of the wave's tiles so far
  float* const bias_tab = st_tab + NWV * 2 * BNW;                      // [BN] bias of the column tile
  u32x4* const src_tab = (u32x4*)(bias_tab + BN);                      // [chunk]{pointer of the chunk's first channel, bytes per pixel}
  float* const aff_tab = (float*)(src_tab + KSMI_MAX_CHUNKS);          // AFF: [chunk][k-group]{scale[8], shift[8]}

  // ---- tile-invariant tables ------------------------------------------------------------------------------------------
  // pixel fragments: the k-group slot of a halo pixel is swizzled by its COLUMN hx only, so the address of tap (ky, kx) is the
  // kx entry plus ky row pitches (12 table registers instead of 36)
  int a_addr[4][KW];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    int p = wm * 64 + mf * 16 + l15;
    if (p >= P) p = 0;
    const int ly = dTW.div(p), lx = p - ly * TW;
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) a_addr[mf][kx] = (ly * HW + lx + kx) * 64 + ((g ^ swz_h(lx + kx)) << 4);
  }
  const int pitch = (DBG && (dbg & 8)) ? 0 : HW * 64;
  int b_addr[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = wn * BNW + nf * 16 + l15;
    b_addr[nf] = n * 64 + ((g ^ swz(n)) << 4);
  }
  if (DBG && (dbg & 8)) {                                              // every lane reads one address (broadcast): LDS bandwidth out of the picture
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int kx = 0; kx < KW; ++kx) a_addr[mf][kx] = 0;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) b_addr[nf] = 0;
  }
  // weight DMA: piece p = wave + 8k covers LDS rows 16p .. 16p+15 of the step's [3 taps][BN] slab; LDS row n of a tap holds output
  // channel 32*(n/32) + 8*((n>>2)&3) + 4*((n>>4)&1) + (n&3) of the column tile (the lean epilogue's ownership, igemm_epilogue.h)
  // (every wave issues WK pieces so that all wait counts are compile-time constants: the pieces beyond the slab -- BN = 64: the
  // second piece of waves 4-7 -- read the zero page into a pad behind the ring)
  int w_off[WK];
  constexpr int nW = WK;
#pragma unroll
  for (int k = 0; k < WK; ++k) {
    const int p = wave + NWV * k;
    const int row = p * 16 + (lane >> 2), sl = lane & 3;
    const int tl = row / BN, n = row - tl * BN;
    const int j = n & 31;
    const int chn = n0 + (n & ~31) + ((j >> 2) & 3) * 8 + (j >> 4) * 4 + (j & 3);
    w_off[k] = (p < WPIECES && chn < d.Npad) ? ((tl * d.Npad + chn) * 64 + ((sl ^ swz(n)) << 4)) : -1;
  }
  // halo DMA: vector v = tid + 512k <-> (halo pixel v >> 2, 16-byte slot v & 3); the slot holds k-group (v & 3) ^ swz_h(hx)
  int h_qb[NHM];
#pragma unroll
  for (int k = 0; k < NHM; ++k) {
    const int pix = (tid + NT * k) >> 2;
    const int hy = dHW.div(pix), hx = pix - hy * HW;
    h_qb[k] = (((tid + NT * k) & 3) ^ swz_h(hx)) << 4;
  }
  // the per-chunk source scalars (virtual concat) go to LDS once: fetched from the kernarg tables inside the K loop they are two
  // dependent scalar loads in front of every halo refill
  for (int ch = tid; ch < nch && ch < KSMI_MAX_CHUNKS; ch += NT) {
    const ksmi_src& sr = d.src[chunk_src_of(d, ch)];
    const uint64_t sp = (uint64_t)(uintptr_t)((const T*)sr.ptr + sr.c_off + chunk_c0_of(d, ch));
    src_tab[ch] = (u32x4){(uint32_t)sp, (uint32_t)(sp >> 32), (uint32_t)sr.C * 2u, 0u};
  }
  if constexpr (AFF) {
    for (int i = tid; i < nch * 32; i += NT) {
      const int ch = i >> 5, j = i & 31;
      const int c0 = chunk_c0_of(d, ch);
      aff_tab[(ch * 4 + (j >> 3)) * 16 + (j & 7)] = d.src[0].scale[c0 + j];
      aff_tab[(ch * 4 + (j >> 3)) * 16 + 8 + (j & 7)] = d.src[0].shift[c0 + j];
    }
  }
  const bool aff_relu = AFF && d.src[0].relu != 0;

  auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
    const int q1 = dTX.div(t);
    const int tx = t - q1 * tilesX;
    b = dTY.div(q1);
    const int ty = q1 - b * tilesY;
    oy0 = ty * TH; ox0 = tx * TW;
  };
  // source pixel index of the lane's halo vectors for tile t (-1: padding / no such tile)
  auto tile_goff = [&](int t, int (&go)[NHM]) {
    int b = 0, oy0 = 0, ox0 = 0;
    const bool live = t < ka.tiles;
    if (live) tile_origin(t, b, oy0, ox0);
    int tv = tid;
    asm volatile("" : "+v"(tv));          // opaque per call: keeps hipcc from hoisting the (tile-invariant) halo coordinates into registers
#pragma unroll
    for (int k = 0; k < NHM; ++k) {
      const int pix = (tv + NT * k) >> 2;
      const int hy = dHW.div(pix), hx = pix - hy * HW;
      // (3 x 3: pad = pad_x = 1, the geometry checks it -- compile-time there: the run-time form of the padding and of the strided
      // views below made every 3 x 3 instance 2-3 % longer on the SNUNet shapes, same box, single stream)
      const int iy = oy0 - (KH == 2 ? d.pad : 1) + hy, ix = ox0 - (KH == 2 ? d.pad_x : 1) + hx;
      const bool ok = live && pix < HP && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
      // dense source, or (2 x 2 phase convolutions) a strided view of a [B, in_H, in_W, C] tensor (ksmi_conv_desc.in_sy ...)
      const int gp = (KH != 2 || d.in_sy == 0) ? (b * d.Hin + iy) * d.Win + ix : (b * d.in_H + (iy * d.in_sy + d.in_oy)) * d.in_W + (ix * d.in_sx + d.in_ox);
      go[k] = ok ? gp : -1;
    }
  };
  const uint32_t zlo = (uint32_t)(uintptr_t)ka.zero, zhi = (uint32_t)((uint64_t)(uintptr_t)ka.zero >> 32);
  // e = src_tab[chunk] (read by the caller ahead of the wait in front of the step)
  auto issue_H = [&](const int (&go)[NHM], const u32x4 e, int slot) {
    const uint64_t sp = ((uint64_t)e[1] << 32) | e[0];
#pragma unroll
    for (int k = 0; k < NHM; ++k) {
      // (two 32-bit selects: a pointer select compiles to two exec-masked DMA instructions)
      const uint64_t av = sp + (uint64_t)(uint32_t)go[k] * e[2] + (uint64_t)(uint32_t)h_qb[k];
      const bool ok = go[k] >= 0;
      const uint32_t lo = ok ? (uint32_t)av : zlo, hi = ok ? (uint32_t)(av >> 32) : zhi;
      glds16_flat((const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo), lds0 + (unsigned)(slot * HSLOT + (k * NT + wave * 64) * 16));
    }
  };
  const unsigned char* const wpk = (const unsigned char*)d.wpk;
  auto issue_W = [&](int ch, int r, int slot) {
    const unsigned char* base = wpk + (size_t)(ch * (KH * KW) + KW * r) * (size_t)d.Npad * 64;
#pragma unroll
    for (int k = 0; k < WK; ++k) {
      const uint64_t av = (uint64_t)(uintptr_t)base + (uint64_t)(uint32_t)w_off[k];
      const bool ok = w_off[k] >= 0;
      const uint32_t lo = ok ? (uint32_t)av : zlo, hi = ok ? (uint32_t)(av >> 32) : zhi;
      const int p = wave + NWV * k;                                   // (wave-uniform select of the landing address)
      const unsigned dst = p < WPIECES ? (unsigned)(nhs * HSLOT + slot * WSLOT + p * 1024) : (unsigned)(nhs * HSLOT + NWS * WSLOT + (wave & 3) * 1024);
      glds16_flat((const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo), lds0 + dst);
    }
  };
  // AFF: relu(x * scale + shift) over the in-image vectors of a landed halo chunk (padding stays zero)
  auto transform = [&](const int (&go)[NHM], int ch, int slot) {
    unsigned char* const sb = smem + slot * HSLOT;
#pragma unroll
    for (int k = 0; k < NHM; ++k) {
      {
        const u32x4 xv = *(const u32x4*)(sb + (tid + NT * k) * 16);
        const int q = h_qb[k] >> 4;
        const f32x4* tab = (const f32x4*)(aff_tab + (ch * 4 + q) * 16);
        const f32x4 s0 = tab[0], s1 = tab[1], h0 = tab[2], h1 = tab[3];
        float x[8];
        vec_unpack<T>(xv, x);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = x[j] * s0[j] + h0[j]; x[4 + j] = x[4 + j] * s1[j] + h1[j]; }
        if (aff_relu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = __builtin_amdgcn_fmed3f(x[j], 0.f, 3.0e38f);
        }
        if (go[k] >= 0) *(u32x4*)(sb + (tid + NT * k) * 16) = vec_pack<T>(x);
      }
    }
  };

  // ---- epilogue state ---------------------------------------------------------------------------------------------------
  const int dC = d.dst[0].C;
  const bool accum = d.dst[0].accumulate != 0;
  // epilogue extras of the plain variant (ResidualBlock of models/changeformer.py:471-483): v = alpha * (acc + bias) + resid, ReLU
  const float alpha = (EPI == 0 && d.alpha != 0.f) ? d.alpha : 1.f;
  const bool has_resid = EPI == 0 && d.resid != nullptr;
  const bool relu_out = EPI == 0 && d.relu_out != 0;
  // statistics and bias live in LDS (a wave owns its statistics row: plain read-modify-write, fixed order = deterministic)
  for (int i = tid; i < NWV * 2 * BNW; i += NT) st_tab[i] = 0.f;
  if (tid < BN) bias_tab[tid] = (d.bias && n0 + tid < d.N) ? d.bias[n0 + tid] : 0.f;
  const bool want_stats = d.stats != nullptr;
  const bool late = ka.stagger && wave >= NWV / 2;                      // (wave-uniform)
  constexpr int LA = NHS - 1;                                           // halo chunks in flight beyond the current one

  // ---- epilogue of one tile (both schedules) --------------------------------------------------------------------------------
  auto tile_epilogue = [&](int t, f32x4 (&acc)[4][NF]) {
      // ---- epilogue of the tile: lane (g, l15) owns channels nc .. nc+7 of pixel l15 of each 16-pixel row group ---------------
      {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        uint32_t opix[4];
        bool okp[4];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
          const int p = wm * 64 + mf * 16 + l15;
          const int ly = dTW.div(p), lx = p - ly * TW;
          const int oy = oy0 + ly, ox = ox0 + lx;
          okp[mf] = p < P && oy < d.Hout && ox < d.Wout;
          // dense destination, or (2 x 2 forward phases) position (oy * out_sy + out_oy, ox * out_sx + out_ox) of a [B, out_H, out_W, C] tensor
          opix[mf] = (KH != 2 || d.out_sy == 0) ? (uint32_t)((b * d.Hout + oy) * d.Wout + ox)
                                                : (uint32_t)((b * d.out_H + (oy * d.out_sy + d.out_oy)) * d.out_W + (ox * d.out_sx + d.out_ox));
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int nc = n0 + (wn * NG + gi) * 32 + g * 8;
          const bool nv = nc < d.N;
          T* const obase = (T*)d.dst[0].ptr + d.dst[0].c_off + nc;
          const T* const mbase = (const T*)d.mask_src + nc;
          float bias8[8], ssum[8], ssq[8];
          {
            const f32x4 a = *(const f32x4*)(bias_tab + (wn * NG + gi) * 32 + g * 8), c = *(const f32x4*)(bias_tab + (wn * NG + gi) * 32 + g * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { bias8[j] = a[j]; bias8[4 + j] = c[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
          }
          float mm[(MASK || GATE) ? 8 : 1], mr[(MASK || GATE) ? 8 : 1], mg[MASK ? 8 : 1], mb[MASK ? 8 : 1];
          if constexpr (MASK || GATE) {
            auto ld8 = [&](const float* qp, float* o, int at) {
              const f32x4 a = *(const f32x4*)(qp + at), c = *(const f32x4*)(qp + at + 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = c[j]; }
            };
#pragma unroll
            for (int j = 0; j < 8; ++j) { mm[j] = 0.f; mr[j] = 0.f; }
            if constexpr (MASK) {
#pragma unroll
              for (int j = 0; j < 8; ++j) { mg[j] = 0.f; mb[j] = 0.f; }
              if (nv) { ld8(d.m_mean, mm, nc); ld8(d.m_rstd, mr, nc); ld8(d.m_scale, mg, nc); ld8(d.m_shift, mb, nc); }
            } else {
              if (nv) { ld8(d.g_mean, mm, nc); ld8(d.g_rstd, mr, nc); }
            }
          }
          // two pixel fragments at a time: the loads of a pair are in flight together (mask / gate / old destination: up to 3 x 16 B)
#pragma unroll
          for (int mh = 0; mh < 4; mh += 2) {
            u32x4 mv[2], zv[2], ov[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int mf = mh + u;
              mv[u] = (u32x4){0u, 0u, 0u, 0u}; zv[u] = mv[u]; ov[u] = mv[u];
              if constexpr (MASK) { if (nv && okp[mf]) mv[u] = *(const u32x4*)(mbase + (size_t)opix[mf] * d.N); }
              if constexpr (EPI == 0) { if (has_resid && nv && okp[mf]) mv[u] = *(const u32x4*)((const T*)d.resid + nc + (size_t)opix[mf] * d.residC); }
              if constexpr (GATE) {
                if (nv && okp[mf]) {
                  mv[u] = *(const u32x4*)((const T*)d.gate_src + nc + (size_t)opix[mf] * d.N);
                  zv[u] = *(const u32x4*)((const T*)d.xhat_src + nc + (size_t)opix[mf] * d.N);
                }
              }
              if (accum && nv && okp[mf]) ov[u] = *(const u32x4*)(obase + (size_t)opix[mf] * dC);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int mf = mh + u;
              const bool ok = nv && okp[mf];
              float v[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) { v[r] = acc[mf][2 * gi][r] + bias8[r]; v[4 + r] = acc[mf][2 * gi + 1][r] + bias8[4 + r]; }
              if constexpr (EPI == 0) {
                float rs[8];
                vec_unpack<T>(mv[u], rs);                                // (zeros without a residual)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  v[j] = v[j] * alpha + rs[j];
                  if (relu_out) v[j] = fmaxf(v[j], 0.f);
                }
              }
              if constexpr (GATE) {
                // total gradient first (the other producers of d out wrote before this launch), then the gate and the BN2 sums
                float o[8], m[8], z[8];
                vec_unpack<T>(ov[u], o); vec_unpack<T>(mv[u], m); vec_unpack<T>(zv[u], z);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  if (accum) v[j] += o[j];
                  if (!(m[j] > 0.f)) v[j] = 0.f;
                  // statistics of the value the later passes READ (the stored bf16), like bn_bwd_apply_add does for its sums
                }
                const u32x4 pk = vec_pack<T>(v);
                float vr[8];
                vec_unpack<T>(pk, vr);
#pragma unroll
                for (int j = 0; j < 8; ++j) if (ok) { ssum[j] += vr[j]; ssq[j] += vr[j] * ((z[j] - mm[j]) * mr[j]); }
                if (ok && !(DBG && (dbg & 4))) *(u32x4*)(obase + (size_t)opix[mf] * dC) = pk;
              } else {
                if constexpr (MASK) {
                  float m[8];
                  vec_unpack<T>(mv[u], m);
#pragma unroll
                  for (int j = 0; j < 8; ++j) {
                    const float xh = (m[j] - mm[j]) * mr[j];
                    if (!(m[j] * mg[j] + mb[j] > 0.f)) v[j] = 0.f;
                    const float q = ElemTraits<T>::cvt(v[j]);          // (statistics of the stored value: see the plain branch)
                    if (ok) { ssum[j] += q; ssq[j] += q * xh; }
                  }
                } else {
                  // statistics of the value the BatchNorm passes READ (the stored bf16), as the reference's BatchNorm sees the stored
                  // tensor; sums of the fp32 accumulators describe a tensor nobody normalises (round 5; the gate branch always did)
                  float q[8];
                  vec_unpack<T>(vec_pack<T>(v), q);
#pragma unroll
                  for (int j = 0; j < 8; ++j) if (ok) { ssum[j] += q[j]; ssq[j] += q[j] * q[j]; }
                }
                if (accum) {
                  float o[8];
                  vec_unpack<T>(ov[u], o);
#pragma unroll
                  for (int j = 0; j < 8; ++j) v[j] += o[j];
                }
                if (ok && !(DBG && (dbg & 4))) *(u32x4*)(obase + (size_t)opix[mf] * dC) = vec_pack<T>(v);
              }
            }
          }
          if (want_stats) {                                             // fold the tile into the wave's statistics row
            float* row = st_tab + (size_t)(wave * 2) * BNW + gi * 32 + g * 8;
            // (all 16 old values are read before the first is written back: one LDS round trip instead of sixteen)
            const f32x4 o0 = *(const f32x4*)row, o1 = *(const f32x4*)(row + 4), o2 = *(const f32x4*)(row + BNW), o3 = *(const f32x4*)(row + BNW + 4);
            f32x4 n0v, n1v, n2v, n3v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              n0v[j] = o0[j] + row16_sum(ssum[j]); n1v[j] = o1[j] + row16_sum(ssum[4 + j]);
              n2v[j] = o2[j] + row16_sum(ssq[j]); n3v[j] = o3[j] + row16_sum(ssq[4 + j]);
            }
            if (l15 == 0) { *(f32x4*)row = n0v; *(f32x4*)(row + 4) = n1v; *(f32x4*)(row + BNW) = n2v; *(f32x4*)(row + BNW + 4) = n3v; }
          }
        }
      }
  };

  // The persistent tile loop, instantiated per DMA role (LATE = the wave refills the rings after the first tap of a step instead of
  // at its top: the two waves of a SIMD then do not queue their DMA issue and their MFMAs at the same moments); one uniform branch
  // per workgroup life instead of one per step.
  auto run_tiles = [&](auto late_tag) {
    constexpr bool LATE = decltype(late_tag)::value;
    int t = pxw;
    int go_c[NHM], go_n[NHM];
    tile_goff(t, go_c);
    tile_goff(t + ka.gx, go_n);
    __syncthreads();                                                  // tables visible; nothing in flight yet
    // ---- prologue: the first LA halo chunks and weight steps 0, 1 of the first tile --------------------------------------------
    issue_H(go_c, src_tab[0], 0);
    if constexpr (LA > 1) issue_H(go_c, src_tab[1], 1);
    issue_W(0, 0, 0);
    issue_W(0, 1, 1);
    if constexpr (AFF) {
      vm_wait_c<(LA > 1 ? nh : 0) + 2 * nW>();                        // chunk 0 landed (this wave's pieces) ...
      lds_barrier();                                                  // ... and everybody's
      transform(go_c, 0, 0);
    }
    int hs = 0;                                                       // halo slot of the current chunk
    STAMP();
    if (DBG && (dbg & 64)) { vm_wait_c<0>(); return; }                // (profiling: setup + ring prologue only)
    for (;;) {
      f32x4 acc[4][NF];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nch; ++c) {
        const int c1 = c + 1 == nch ? 0 : c + 1;                      // chunk of the next two weight steps after this chunk's rows
        const unsigned char* lds_h = smem + hs * HSLOT;
        const int hprev = hs == 0 ? NHS - 1 : hs - 1;                 // slot of chunk - 1 = slot of chunk + LA
        const int hnext = hs + 1 == NHS ? 0 : hs + 1;
        // source scalars and pixel offsets of the halo chunk issued in step r = 0 (LA chunks ahead: the next tile's near the end)
        int c2 = c + LA;
        const bool nextt = c2 >= nch;
        if (nextt) c2 -= nch;
        const u32x4 ent = src_tab[c2];
        int go_i[NHM];
#pragma unroll
        for (int k = 0; k < NHM; ++k) go_i[k] = nextt ? go_n[k] : go_c[k];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          // the weights of this step (and every older DMA) landed; AFF with one halo chunk of lookahead transforms the NEXT chunk
          // in step r = 2: that chunk was issued after this step's weights, so only the youngest nW pieces may be outstanding
          if constexpr (AFF && LA == 1) { if (r == 2) vm_wait_c<nW>(); else if (r == 1) vm_wait_c<nW + nh>(); else vm_wait_c<nW>(); }
          else { if (r == 0) vm_wait_c<nW>(); else vm_wait_c<nW + nh>(); }
          if (!(DBG && (dbg & 16))) lds_barrier();                    // ... for every wave; the slots refilled below are free
          auto refill = [&]() {
            if (DBG && (dbg & 2)) return;
            if (r == 0) {
              issue_W(c, 2, 2);
              issue_H(go_i, ent, hprev);
            } else if (r == 1) {
              issue_W(c1, 0, 0);
            } else {
              issue_W(c1, 1, 1);
            }
          };
          if constexpr (!LATE) refill();
          if constexpr (AFF) {
            if (r == 2) {                                             // next chunk: landed (see the wait above), published by the next barrier
              if (c + 1 == nch) transform(go_n, 0, hnext); else transform(go_c, c + 1, hnext);
            }
          }
          const unsigned char* lds_w = wring + r * WSLOT;
          if (!(DBG && (dbg & 1))) {
            // fragments of tap j+1 are requested before the MFMAs of tap j
            u32x4 fa[2][4], fb[2][NF];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) fa[0][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][0]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) fb[0][nf] = *(const u32x4*)(lds_w + b_addr[nf]);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const int cur = j & 1, nxt = cur ^ 1;
              if (j + 1 < 3) {
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) fa[nxt][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][j + 1]);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) fb[nxt][nf] = *(const u32x4*)(lds_w + (j + 1) * BN * 64 + b_addr[nf]);
              }
#pragma unroll
              for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) mma16<T>(acc[mf][nf], fb[cur][nf], fa[cur][mf]);   // D = W * X^T
              if constexpr (LATE) { if (j == 0) refill(); }
            }
          } else if constexpr (LATE) refill();
        }
        hs = hnext;
      }
      STAMP();
      tile_epilogue(t, acc);
      STAMP();
      if (t + ka.gx >= ka.tiles) break;
      t += ka.gx;
#pragma unroll
      for (int k = 0; k < NHM; ++k) go_c[k] = go_n[k];
      tile_goff(t + ka.gx, go_n);
    }
    vm_wait_c<0>();                                                   // the ring's tail (zero-page reads) has landed: LDS is ours again
  };
  // ROTATED schedule (ka.rot): the step barrier sits between the MFMAs of tap 1 and tap 2 of the PREVIOUS step.  The fragments of that
  // tap 2 are read before the barrier (their slots may then be refilled), so a wave leaves the barrier with 4 * NF MFMAs to issue at
  // once; they cover the LDS latency of the new step's first fragments and the DMA issue of the refill, which sit right behind them.
  // Same data flow, slots and wait counts as above; the two fragment register sets swap roles every step (3 steps per chunk: the
  // chunk body exists for both parities).
  auto run_tiles_rot = [&](auto) {
    int t = pxw;
    int go_c[NHM], go_n[NHM];
    tile_goff(t, go_c);
    tile_goff(t + ka.gx, go_n);
    __syncthreads();
    issue_H(go_c, src_tab[0], 0);
    if constexpr (LA > 1) issue_H(go_c, src_tab[1], 1);
    issue_W(0, 0, 0);
    issue_W(0, 1, 1);
    if constexpr (AFF) {
      vm_wait_c<(LA > 1 ? nh : 0) + 2 * nW>();
      lds_barrier();
      transform(go_c, 0, 0);
    }
    int hs = 0;
    for (;;) {
      f32x4 acc[4][NF];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // the two fragment register sets live inside a tile only (nothing is held across the epilogue)
      u32x4 fa[2][4], fb[2][NF];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) fa[u][mf] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) fb[u][nf] = (u32x4){0u, 0u, 0u, 0u};
      }
      bool have = false;                                              // a held tap 2 of the previous step (not at the first step of a tile)
      auto mma_set = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) mma16<T>(acc[mf][nf], fb[S][nf], fa[S][mf]);
      };
      auto chunk = [&](auto par_tag, int c) {
        constexpr int PAR = decltype(par_tag)::value;
        const int c1 = c + 1 == nch ? 0 : c + 1;
        const unsigned char* lds_h = smem + hs * HSLOT;
        const int hprev = hs == 0 ? NHS - 1 : hs - 1;
        const int hnext = hs + 1 == NHS ? 0 : hs + 1;
        int c2 = c + LA;
        const bool nextt = c2 >= nch;
        if (nextt) c2 -= nch;
        const u32x4 ent = src_tab[c2];
        int go_i[NHM];
#pragma unroll
        for (int k = 0; k < NHM; ++k) go_i[k] = nextt ? go_n[k] : go_c[k];
        auto step = [&](auto r_tag) {
          constexpr int r = decltype(r_tag)::value;
          constexpr int Hs = (PAR + r) & 1, Os = Hs ^ 1;               // held set (tap 2 of the previous step) / the other one
          if constexpr (AFF && LA == 1) { if (r == 2) vm_wait_c<nW>(); else if (r == 1) vm_wait_c<nW + nh>(); else vm_wait_c<nW>(); }
          else { if (r == 0) vm_wait_c<nW>(); else vm_wait_c<nW + nh>(); }
          if (!(DBG && (dbg & 16))) lds_barrier();
          const unsigned char* lds_w = wring + r * WSLOT;
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) fa[Os][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][0]);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) fb[Os][nf] = *(const u32x4*)(lds_w + b_addr[nf]);
          if (have) mma_set(std::integral_constant<int, Hs>{});
          __builtin_amdgcn_sched_barrier(0);                            // (the reads of tap 1 below reuse set Hs: keep them behind these MFMAs instead of a third register set)
          if (!(DBG && (dbg & 2))) {
            if (r == 0) { issue_W(c, 2, 2); issue_H(go_i, ent, hprev); }
            else if (r == 1) issue_W(c1, 0, 0);
            else issue_W(c1, 1, 1);
          }
          if constexpr (AFF) {
            if (r == 2) { if (c + 1 == nch) transform(go_n, 0, hnext); else transform(go_c, c + 1, hnext); }
          }
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) fa[Hs][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][1]);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) fb[Hs][nf] = *(const u32x4*)(lds_w + 1 * BN * 64 + b_addr[nf]);
          mma_set(std::integral_constant<int, Os>{});      // tap 0
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) fa[Os][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][2]);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) fb[Os][nf] = *(const u32x4*)(lds_w + 2 * BN * 64 + b_addr[nf]);
          mma_set(std::integral_constant<int, Hs>{});      // tap 1; set Os now holds tap 2
          __builtin_amdgcn_sched_barrier(0);
          have = true;
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        hs = hnext;
      };
      // three steps per chunk: the held set changes sides every chunk -> chunk pairs as straight-line code (no parity to merge)
      int c = 0;
      for (; c + 1 < nch; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        chunk(std::integral_constant<int, 1>{}, c + 1);
      }
      // the tap 2 of the tile's last step: held in set 0 after an even number of chunks, in set 1 after the odd tail chunk
      if (c < nch) {
        chunk(std::integral_constant<int, 0>{}, c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mma_set(std::integral_constant<int, 1>{});
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mma_set(std::integral_constant<int, 0>{});
      }
      STAMP();
      tile_epilogue(t, acc);
      STAMP();
      if (t + ka.gx >= ka.tiles) break;
      t += ka.gx;
#pragma unroll
      for (int k = 0; k < NHM; ++k) go_c[k] = go_n[k];
      tile_goff(t + ka.gx, go_n);
    }
    vm_wait_c<0>();
  };

  // DEEP schedule (see the constant): the rotated step body of run_tiles_rot with
  //   * weight slot = global step index mod 5; at step s the weights of step s + 4 are issued into the slot step s - 1 read;
  //   * a kernel-row-0 step issues its weights first, then the halo of chunk + 2 (as before: the slot of chunk - 1);
  //   * waits: the DMA instructions younger than the weights of the step are, in issue order,
  //       row 0: W(s+1) W(s+2) W(s+3) + the halo issued one chunk ago            -> vmcnt(3 nW + nh)
  //       row 1: the halo issued in row 0 of the PREVIOUS chunk sits behind these weights as well -> vmcnt(3 nW + 2 nh)
  //       row 2:                                                                  -> vmcnt(3 nW + nh)
  //     so a halo chunk issued in step (c, 0) is forced by the wait of step (c + 1, 2): five steps of lead instead of three.
  //   The prologue issues what the virtual steps -4 .. -1 would have (H0 | W0 | W1 H1 | W2 | W3).
  auto run_tiles_deep = [&](auto) {
    int t = pxw;
    int go_c[NHM], go_n[NHM];
    tile_goff(t, go_c);
    tile_goff(t + ka.gx, go_n);
    __syncthreads();
    issue_H(go_c, src_tab[0], 0);
    issue_W(0, 0, 0);
    {
      const int c1 = nch > 1 ? 1 : 0;                                 // (a one-chunk tile re-reads chunk 0: the geometry needs nch >= 2 anyway)
      issue_W(0, 1, 1);
      issue_H(go_c, src_tab[c1], 1);
      issue_W(0, 2, 2);
      issue_W(c1, 0, 3);
    }
    int hs = 0;
    int ws = 0;                                                       // weight slot of the current step
    int ic = nch > 1 ? 1 : 0, ir = 1;                                 // (chunk, row) of the step whose weights are issued next: step 4
    for (;;) {
      f32x4 acc[4][NF];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      u32x4 fa[2][4], fb[2][NF];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) fa[u][mf] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) fb[u][nf] = (u32x4){0u, 0u, 0u, 0u};
      }
      bool have = false;
      auto mma_set = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) mma16<T>(acc[mf][nf], fb[S][nf], fa[S][mf]);
      };
      auto chunk = [&](auto par_tag, int c) {
        constexpr int PAR = decltype(par_tag)::value;
        const unsigned char* lds_h = smem + hs * HSLOT;
        const int hprev = hs == 0 ? NHS - 1 : hs - 1;
        const int hnext = hs + 1 == NHS ? 0 : hs + 1;
        int c2 = c + LA;
        const bool nextt = c2 >= nch;
        if (nextt) c2 -= nch;
        const u32x4 ent = src_tab[c2];
        int go_i[NHM];
#pragma unroll
        for (int k = 0; k < NHM; ++k) go_i[k] = nextt ? go_n[k] : go_c[k];
        auto step = [&](auto r_tag) {
          constexpr int r = decltype(r_tag)::value;
          constexpr int Hs = (PAR + r) & 1, Os = Hs ^ 1;
          if constexpr (r == 1) vm_wait_c<3 * nW + 2 * nh>(); else vm_wait_c<3 * nW + nh>();
          lds_barrier();
          const unsigned char* lds_w = wring + ws * WSLOT;
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) fa[Os][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][0]);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) fb[Os][nf] = *(const u32x4*)(lds_w + b_addr[nf]);
          if (have) mma_set(std::integral_constant<int, Hs>{});
          __builtin_amdgcn_sched_barrier(0);
          {
            const int wslot = ws == 0 ? NWS - 1 : ws - 1;             // the slot step s - 1 read = (s + 4) mod 5
            issue_W(ic, ir, wslot);
            if (r == 0) issue_H(go_i, ent, hprev);
            if (++ir == 3) { ir = 0; ic = ic + 1 == nch ? 0 : ic + 1; }
          }
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) fa[Hs][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][1]);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) fb[Hs][nf] = *(const u32x4*)(lds_w + 1 * BN * 64 + b_addr[nf]);
          mma_set(std::integral_constant<int, Os>{});      // tap 0
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mf = 0; mf < 4; ++mf) fa[Os][mf] = *(const u32x4*)(lds_h + r * pitch + a_addr[mf][2]);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) fb[Os][nf] = *(const u32x4*)(lds_w + 2 * BN * 64 + b_addr[nf]);
          mma_set(std::integral_constant<int, Hs>{});      // tap 1; set Os now holds tap 2
          __builtin_amdgcn_sched_barrier(0);
          have = true;
          ws = ws + 1 == NWS ? 0 : ws + 1;
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        hs = hnext;
      };
      int c = 0;
      for (; c + 1 < nch; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        chunk(std::integral_constant<int, 1>{}, c + 1);
      }
      if (c < nch) {
        chunk(std::integral_constant<int, 0>{}, c);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mma_set(std::integral_constant<int, 1>{});
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        mma_set(std::integral_constant<int, 0>{});
      }
      tile_epilogue(t, acc);
      if (t + ka.gx >= ka.tiles) break;
      t += ka.gx;
#pragma unroll
      for (int k = 0; k < NHM; ++k) go_c[k] = go_n[k];
      tile_goff(t + ka.gx, go_n);
    }
    vm_wait_c<0>();
  };

  // CHUNK schedule (see the constant).  Per chunk c: wait for W(c) (3 rows) and H(c) -> barrier -> issue W(c + 1) into the other half of
  // the weight ring and the halo of chunk c + LA into the slot chunk c - 1 used -> 9 taps.  Issue order per chunk: weights, then halo:
  //   LA = 2: the halo issued one chunk ago (for c + 1) is the only DMA younger than W(c)  -> vmcnt(nh);   LA = 1: nothing is -> vmcnt(0).
  auto run_tiles_chunk = [&](auto) {
    int t = pxw;
    int go_c[NHM], go_n[NHM];
    tile_goff(t, go_c);
    tile_goff(t + ka.gx, go_n);
    __syncthreads();
    issue_H(go_c, src_tab[0], 0);
#pragma unroll
    for (int r = 0; r < KH; ++r) issue_W(0, r, r);
    if constexpr (LA > 1) issue_H(go_c, src_tab[1], 1);
    int hs = 0;
    int wp = 0;                                                       // half of the weight ring that holds the current chunk
    for (;;) {
      f32x4 acc[4][NF];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nch; ++c) {
        const int c1 = c + 1 == nch ? 0 : c + 1;
        const unsigned char* lds_h = smem + hs * HSLOT;
        const int hprev = hs == 0 ? NHS - 1 : hs - 1;
        const int hnext = hs + 1 == NHS ? 0 : hs + 1;
        int c2 = c + LA;
        const bool nextt = c2 >= nch;
        if (nextt) c2 -= nch;
        const u32x4 ent = src_tab[c2];
        int go_i[NHM];
#pragma unroll
        for (int k = 0; k < NHM; ++k) go_i[k] = nextt ? go_n[k] : go_c[k];
        if constexpr (LA > 1) vm_wait_c<nh>(); else vm_wait_c<0>();
        lds_barrier();
        const unsigned char* lds_w = wring + wp * KH * WSLOT;
        u32x4 fa[2][4], fb[2][NF];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) fa[0][mf] = *(const u32x4*)(lds_h + a_addr[mf][0]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) fb[0][nf] = *(const u32x4*)(lds_w + b_addr[nf]);
        {                                                             // refill behind the first fragment requests (their latency covers the address arithmetic)
          const int wn_ = (wp ^ 1) * KH;
#pragma unroll
          for (int r = 0; r < KH; ++r) issue_W(c1, r, wn_ + r);
          issue_H(go_i, ent, LA > 1 ? hprev : hnext);
        }
#pragma unroll
        for (int j = 0; j < KH * KW; ++j) {
          const int cur = j & 1, nxt = cur ^ 1;
          if (j + 1 < KH * KW) {
            const int r1 = (j + 1) / KW, k1 = (j + 1) % KW;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) fa[nxt][mf] = *(const u32x4*)(lds_h + r1 * pitch + a_addr[mf][k1]);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) fb[nxt][nf] = *(const u32x4*)(lds_w + (r1 * KW + k1) * BN * 64 + b_addr[nf]);
          }
#pragma unroll
          for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma16<T>(acc[mf][nf], fb[cur][nf], fa[cur][mf]);
          __builtin_amdgcn_sched_barrier(0);
        }
        hs = hnext;
        wp ^= 1;
      }
      tile_epilogue(t, acc);
      if (t + ka.gx >= ka.tiles) break;
      t += ka.gx;
#pragma unroll
      for (int k = 0; k < NHM; ++k) go_c[k] = go_n[k];
      tile_goff(t + ka.gx, go_n);
    }
    vm_wait_c<0>();
  };
  STAMP();
  if (DBG && (dbg & 32)) return;                                        // (profiling: table setup only)
  if (pxw < ka.tiles) {
    if constexpr (CHUNK) run_tiles_chunk(0);
    else if constexpr (DEEP) run_tiles_deep(0);
    else if constexpr (ROT != 0) run_tiles_rot(0);
    else { if (late) run_tiles(std::true_type{}); else run_tiles(std::false_type{}); }
  }
  if (DBG && (dbg & 128)) {
    STAMP();
    if (blockIdx.x == 0 && tid == 0 && d.stats) { long long* o = (long long*)d.stats; for (int i = 0; i < 8; ++i) o[i] = i < nstamp ? stamp[i] : 0; }
    return;
  }
  // ---- statistics: one row per pixel-axis workgroup ---------------------------------------------------------------------------
  if (want_stats) {
    __syncthreads();
    if (tid < 2 * BN) {                                               // wave = wn * WM + wm: sum the WM pixel groups of a column group
      const int which = tid / BN, nn = tid - which * BN;
      const int grp = nn / BNW, n = nn - grp * BNW;
      const float* rg = st_tab + (size_t)grp * WM * 2 * BNW;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += rg[(w * 2 + which) * BNW + n];
      if (n0 + nn < d.Npad) d.stats[((size_t)pxw * 2 + which) * d.Npad + n0 + nn] = v;
    }
  }
}

}  // namespace

// ---- host side --------------------------------------------------------------------------------------------------------------
namespace {
// output patch of a workgroup with at most pmax pixels and a halo of at most hmax pixels: fewest tiles, then smallest halo
void ig4_patch(int H, int W, int pmax, int hmax, int* th_o, int* tw_o) {
  double best = 1e30;
  int bth = 1, btw = 1;
  for (int tw = 1; tw <= W && tw <= pmax; ++tw)
    for (int th = 1; th <= H && th * tw <= pmax; ++th) {
      const int hp = (th + 2) * (tw + 2);
      if (hp > hmax) continue;
      const int tiles = ((H + th - 1) / th) * ((W + tw - 1) / tw);
      const double cost = (double)tiles * pmax * (1.0 + 0.15 * hp / pmax);
      if (cost < best) { best = cost; bth = th; btw = tw; }
    }
  *th_o = bth; *tw_o = btw;
}
}  // namespace

bool ksmi_igemm4_geom(const ksmi_conv_desc* d, int dtype, ksmi_igemm4_geom_t* g) {
  static const bool off = ksmi_knob_is_set("KSMI_IGEMM4_OFF");
  if (off || dtype != KSMI_BF16) return false;
  // 2 x 2 stride-1 phase convolutions of ConvTranspose2d(k4, s2, p1) and its input gradient (round 5): strided views, padding 0 | 1 per
  // phase, 128-column tiles, the chunk-granular schedule.  KSMI_IG4_K2=0 sends them back to igemm2.
  static const int k2_on = ksmi_knob_int("KSMI_IG4_K2", 1);
  const bool k2 = d->KH == 2 && d->KW == 2;
  if (k2) {
    if (!k2_on || d->stride != 1 || (unsigned)d->pad > 1u || (unsigned)d->pad_x > 1u) return false;
    if (d->Npad % 128 || d->src[0].scale || d->gate_src || (d->out_sy && (d->mask_src || d->dst[0].accumulate))) return false;
    if (d->alpha != 0.f || d->resid || d->relu_out) return false;
  } else if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->pad_x != 1) return false;
  if (d->Hin != d->Hout || d->Win != d->Wout) return false;
  if (d->nchunks < 2) return false;
  if (d->nchunks > KSMI_MAX_CHUNKS) return false;                   // (source table in LDS)
  if (d->ndst != 1 || d->dst[0].n_begin != 0) return false;
  if (((d->out_sy || d->in_sy) && !k2) || d->ps_cout) return false;
  if (d->out_sy && ((size_t)d->B * d->out_H * d->out_W >= ((size_t)1 << 31))) return false;
  if ((d->alpha != 0.f || d->resid || d->relu_out) && (d->mask_src || d->gate_src)) return false;      // extras: plain epilogue only
  if (d->resid && (((uintptr_t)d->resid & 15) || (d->residC % 8))) return false;
  if ((d->dst[0].C % 8) || (d->dst[0].c_off % 8)) return false;
  // N % 8 != 0 (2- / 3-class heads into a destination with a channel stride of 8: ChangeFormer's change_probability, 256 -> 2): the last
  // 8-channel group is stored whole, its pad channels as zeros (zero weights, no bias beyond N) -- plain epilogue only
  // (KSMI_IG4_THIN=0: back to the tile kernel)
  static const int thin_on = ksmi_knob_int("KSMI_IG4_THIN", 1);
  if ((d->N % 8) && (!thin_on || k2 || d->mask_src || d->gate_src || d->resid || d->dst[0].accumulate || d->alpha != 0.f || d->relu_out ||
                     d->dst[0].c_off + ((d->N + 7) & ~7) > d->dst[0].C)) return false;
  if (d->Npad != 32 && (d->Npad % 64)) return false;
  auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
  if (!al16(d->dst[0].ptr) || !al16(d->bias) || !al16(d->mask_src) || !al16(d->m_mean) || !al16(d->m_rstd) || !al16(d->m_scale) || !al16(d->m_shift))
    return false;
  for (int i = 0; i < d->nsrc; ++i) {
    if (d->src[i].c_len % 32) return false;
    if (d->src[i].scale && (i > 0 || d->nsrc != 1)) return false;
    if (!al16(d->src[i].ptr) || (d->src[i].C % 8) || (d->src[i].c_off % 8)) return false;
    const size_t ipx = d->in_sy ? (size_t)d->B * d->in_H * d->in_W : (size_t)d->B * d->Hin * d->Win;
    if (ipx * (size_t)d->src[i].C * 2 >= ((size_t)1 << 32)) return false;
  }
  if ((size_t)d->B * d->Hout * d->Wout >= ((size_t)1 << 31)) return false;
  const bool aff = d->src[0].scale != nullptr;
  if (aff && (d->mask_src || d->gate_src)) return false;
  if (d->mask_src && d->gate_src) return false;
  if (d->gate_src && (!d->xhat_src || !d->g_mean || !d->g_rstd || !al16(d->gate_src) || !al16(d->xhat_src) || !al16(d->g_mean) || !al16(d->g_rstd)))
    return false;
  if (aff && d->nchunks > 64) return false;                         // affine table: 8 KB
  char v_env[32];                                                   // (a knob: the tests force each variant) "wm,nf"
  int wm_force = 0, nf_force = 0;
  if (ksmi_knob_str("KSMI_IGEMM4_VAR", v_env, sizeof(v_env))) sscanf(v_env, "%d,%d", &wm_force, &nf_force);
  const int cus = ksmi_knob_int("KSMI_IGEMM4_CUS", 256);            // (a knob: the tests shrink the grid to force many rounds)
  // variants (8 waves): WM = 4 pixel groups x 2 column groups of 16 NF columns (256-pixel patch the descriptor chose), or
  // WM = 8 pixel groups x 1 column group (512-pixel patch chosen here): the wider the wave tile, the fewer LDS bytes per MFMA
  struct Var { int wm, nf, nwv; };
  Var cand[5];
  int nc = 0;
  // two 4-wave workgroups per CU for the 32-column layers (KSMI_IG4_NW4: 1 = on, 0 = off; read once)
  static const int nw4 = ksmi_knob_int("KSMI_IG4_NW4", 0);
  if (d->Npad == 32 && nw4 && !k2) cand[nc++] = {4, 2, 4};          // 256 px x 32, wave 64 x 32, two workgroups per CU
  if (d->Npad % 128 == 0) cand[nc++] = {4, 4, 8};                   // 256 px x 128 columns, wave 64 x 64
  // 64 columns exactly (SNUNet level 1): 256 px x 64 tiles make 7 rounds of 224 workgroups at 112^2 x 32 (88 % of the slots filled) where
  // the 512 px ones make 4 rounds of 208 (77 %): 2-8 % shorter per launch (profiles/r05_ig4_variants.txt (e)); KSMI_IG4_N64=84 restores the old order
  static const int n64 = ksmi_knob_int("KSMI_IG4_N64", 42);
  // (2 x 2 phase convolutions: the 256 px x 128 shape is the ONLY instance that exists -- no 64-column candidates for them, so a geometry
  // this function accepts is always launchable; a shape the fill rule below would have skipped simply stays on that one tile)
  if (!k2) {
    if (d->Npad == 64 && n64 == 42) { cand[nc++] = {4, 2, 8}; cand[nc++] = {8, 4, 8}; }
    else if (d->Npad % 64 == 0) { cand[nc++] = {8, 4, 8}; cand[nc++] = {4, 2, 8}; }   // 512 px x 64 (wave 64 x 64) / 256 px x 64 (wave 64 x 32)
    if (d->Npad == 32) cand[nc++] = {8, 2, 8};                        // 512 px x 32, wave 64 x 32
  }
  for (int ci = 0; ci < nc; ++ci) {
    const int wm = cand[ci].wm, nf = cand[ci].nf, nwv = cand[ci].nwv;
    if (wm_force && (wm != wm_force || nf != nf_force)) continue;
    if (k2 && !(wm == 4 && nf == 4 && nwv == 8)) continue;          // (belt and braces: the one 2 x 2 instance, ksmi_igemm4_launch)
    const int bn = (nwv / wm) * 16 * nf;
    int th, tw;
    if (wm == 4) { th = d->TH; tw = d->TW; if (th * tw > 256) continue; }
    else {
      ig4_patch(d->Hout, d->Wout, 512, IG4_NHMAX * 128, &th, &tw);
      char pe[32];                                                  // probes: "th,tw" of the 512-pixel variants (DRAM-locality experiments)
      int pth = 0, ptw = 0;
      if (ksmi_knob_str("KSMI_IG4_PATCH", pe, sizeof(pe)) && sscanf(pe, "%d,%d", &pth, &ptw) == 2 && pth > 0 && ptw > 0 && pth * ptw <= 512 && (pth + 2) * (ptw + 2) <= IG4_NHMAX * 128) { th = pth; tw = ptw; }
    }
    const int hp = (th + d->KH - 1) * (tw + d->KW - 1);
    const int tilesX = (d->Wout + tw - 1) / tw, tilesY = (d->Hout + th - 1) / th;
    const int tiles = d->B * tilesX * tilesY;
    const int gy = d->Npad / bn;
    if (!wm_force && tiles < 64) continue;                          // tiny maps (14 x 14 x batch 32 = 32 patches): igemm2's smaller tiles fill the machine better (measured)
    // the machine has to fill: prefer the wide variants only when they still give (nearly) every CU a workgroup
    if (!wm_force && ci + 1 < nc && (size_t)tiles * gy < 192) continue;
    g->WM = wm; g->NF = nf; g->nwv = nwv; g->th = th; g->tw = tw;
    g->tiles = tiles; g->gy = gy;
    g->nh = nwv == 4 ? 6 : (wm == 4 ? 3 : IG4_NHMAX);               // (= the kernel's NHM)
    g->hslot = g->nh * nwv * 1024;
    if (hp * 64 > g->hslot) continue;
    const size_t tabs = (size_t)(nwv * 2 * 16 * nf + bn) * 4 + (size_t)KSMI_MAX_CHUNKS * 16 + (aff ? (size_t)d->nchunks * 32 * 2 * 4 : 0);
    static const int rot_on = ksmi_knob_int("KSMI_IG4_ROT", 1);
    static const int deep_on = ksmi_knob_int("KSMI_IG4_DEEP", 0);   // (the kernel's DEEP: compiled in; the switch forces the per-role schedule instead)
    // chunk-granular schedule (kernel: CHUNK = ROT == 3) for the NF = 2 tiles: KSMI_IG4_CHUNK (1 = on)
    static const int chunk_on = ksmi_knob_int("KSMI_IG4_CHUNK", 0);
    const bool chunk = k2 || (rot_on && chunk_on && !aff && nwv == 8 && nf == 2);
    const int nhs_ = (nwv == 4 || (wm == 8 && nf == 4) || (chunk && wm == 8)) ? 2 : 3;
    const bool deep = !chunk && rot_on && deep_on && !aff && nhs_ == 3 && nwv == 8 && nf == 2;      // <8,2> (32 columns) and <4,2> (64 columns)
    g->deep = chunk ? 2 : (deep ? 1 : 0);
    const size_t wr = (chunk ? 2 * d->KH : (deep ? 5 : 3)) * (size_t)(d->KW * bn * 64);
    g->nhs = nhs_;                                                  // (= the kernel's NHS)
    g->lds = (size_t)g->nhs * g->hslot + wr + ((d->KW * bn / 16) % nwv ? 4096 : 0) + tabs;
    const int per_cu = nwv == 4 ? 2 : 1;
    if (g->lds * per_cu > 160 * 1024) continue;
    int gx = cus * per_cu / gy;
    if (gx < 1) gx = 1;
    if (gx > tiles) gx = tiles;
    const int rounds = (tiles + gx - 1) / gx;                       // equalise: the smallest grid with the same number of rounds
    gx = (tiles + rounds - 1) / rounds;
    if (gx >= 8 && ((gx + 7) / 8 * 8) * gy <= (cus > 8 ? cus : 8) * per_cu) gx = (gx + 7) / 8 * 8;   // XCD-aware placement wants a multiple of 8
    g->gx = gx;
    return true;
  }
  return false;
}

// dry run (ksmi_igemm4_launchable): walk the instance selection of ksmi_igemm4_launch without touching the device
static thread_local bool ig4_dry = false;
#define KSMI_IG4_DRY_RETURN do { if (ig4_dry) return 0; } while (0)

bool ksmi_igemm4_launchable(const ksmi_conv_desc* d, const ksmi_igemm4_geom_t* g) {
  ig4_dry = true;
  const int rc = ksmi_igemm4_launch(d, g, nullptr);
  ig4_dry = false;
  return rc == 0;
}

int ksmi_igemm4_launch(const ksmi_conv_desc* d, const ksmi_igemm4_geom_t* g, hipStream_t st) {
  Ig4Args ka;
  ka.d = *d;
  ka.th = g->th; ka.tw = g->tw;
  const int tilesX = (d->Wout + g->tw - 1) / g->tw, tilesY = (d->Hout + g->th - 1) / g->th;
  ka.m_tw = fastdiv_magic(g->tw); ka.m_hw = fastdiv_magic(g->tw + d->KW - 1); ka.m_tx = fastdiv_magic(tilesX); ka.m_ty = fastdiv_magic(tilesY);
  ka.dbg = ksmi_knob_int("KSMI_IG4_DBG", 0);
  static const int stag = ksmi_knob_int("KSMI_IG4_STAGGER", 1);
  ka.stagger = stag;
  static const int rot = ksmi_knob_int("KSMI_IG4_ROT", 1);      // (rotated schedule: +3-7 % on the long-K shapes)
  ka.rot = rot;
  ka.hslot = g->hslot; ka.nh = g->nh; ka.nhs = g->nhs;
  ka.tiles = g->tiles; ka.gx = g->gx; ka.gy = g->gy;
  const unsigned char* const zero_page = ig4_dry ? nullptr : ig4_zero();
  if (!zero_page && !ig4_dry) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm4: zero page");
  ka.zero = zero_page;
  const dim3 grid(g->gx * g->gy);
  const bool aff = d->src[0].scale != nullptr, mask = d->mask_src != nullptr, gate = d->gate_src != nullptr;
#define KSMI_G4(WM_, NF_, AFF_, MASK_)                                                               \
  do {                                                                                               \
    auto kfn = igemm4_kernel<WM_, NF_, AFF_, MASK_>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                                                 \
    static bool attr_set = false;        /* (one driver call per instantiation, not per launch) */  \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
#define KSMI_G4D(WM_, NF_)                                                                           \
  do {                                                                                               \
    auto kfn = igemm4_kernel<WM_, NF_, false, 0, false, 1>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                                          \
    static bool attr_set = false;                                                                    \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
#define KSMI_G4R(WM_, NF_, AFF_, EPI_)                                                               \
  do {                                                                                               \
    auto kfn = igemm4_kernel<WM_, NF_, AFF_, EPI_, false, 0, 1>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                                     \
    static bool attr_set = false;                                                                    \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
#define KSMI_G4V(WM_, NF_)                                                                           \
  do { if (ka.rot) { if (aff) KSMI_G4R(WM_, NF_, true, 0); else if (mask) KSMI_G4R(WM_, NF_, false, 1); else if (gate) KSMI_G4R(WM_, NF_, false, 2); else KSMI_G4R(WM_, NF_, false, 0); } \
       if (aff) KSMI_G4(WM_, NF_, true, 0); else if (mask) KSMI_G4(WM_, NF_, false, 1); else if (gate) KSMI_G4(WM_, NF_, false, 2);   \
       else if (d->dir == 1) KSMI_G4D(WM_, NF_); else KSMI_G4(WM_, NF_, false, 0); } while (0)
  if (ka.dbg && !aff && !mask && !gate && g->NF == 2 && g->WM == 8 && g->nwv == 8) {      // (the 32-column level-0 shape)
    auto kfn = igemm4_kernel<8, 2, false, 0, true>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);
    return ksmi_check_launch("igemm4");
  }
  if (ka.dbg && !aff && !mask && !gate && g->NF == 4 && g->nwv == 8) {        // profiling switches: separate instantiations of the plain kernels
    if (g->WM == 4) {
      auto kfn = igemm4_kernel<4, 4, false, 0, true>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);
    } else {
      auto kfn = igemm4_kernel<8, 4, false, 0, true>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);
    }
    return ksmi_check_launch("igemm4");
  }
  if (g->nwv == 4) {
#define KSMI_G4N(AFF_, EPI_, ROT_)                                                                   \
  do {                                                                                               \
    auto kfn = igemm4_kernel<4, 2, AFF_, EPI_, false, 0, ROT_ ? 1 : 0, 4>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                           \
    static bool attr_set = false;                                                                    \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(256), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
    if (g->WM != 4 || g->NF != 2) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm4: no 4-wave instance");
    if (ka.rot) { if (aff) KSMI_G4N(true, 0, true); else if (mask) KSMI_G4N(false, 1, true); else if (gate) KSMI_G4N(false, 2, true); else KSMI_G4N(false, 0, true); }
    if (aff) KSMI_G4N(true, 0, false); else if (mask) KSMI_G4N(false, 1, false); else if (gate) KSMI_G4N(false, 2, false); else KSMI_G4N(false, 0, false);
#undef KSMI_G4N
  }
  if (g->deep == 2 && d->KH == 2) {                                 // the 2 x 2 phase convolutions: 256 px x 128 columns, chunk schedule
#define KSMI_G4K2(EPI_)                                                                              \
  do {                                                                                               \
    auto kfn = igemm4_kernel<4, 4, false, EPI_, false, 0, 3, 8, 2, 2>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                               \
    static bool attr_set = false;                                                                    \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
    if (aff || gate || g->nwv != 8 || g->WM != 4 || g->NF != 4) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm4: no 2 x 2 instance");
    if (mask) KSMI_G4K2(1); else KSMI_G4K2(0);
#undef KSMI_G4K2
  }
  if (g->deep == 2) {
#define KSMI_G4CH(WM_, NF_, EPI_)                                                                    \
  do {                                                                                               \
    auto kfn = igemm4_kernel<WM_, NF_, false, EPI_, false, 0, 3>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                                    \
    static bool attr_set = false;                                                                    \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
    if (aff || g->nwv != 8 || g->NF != 2) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm4: no chunk-schedule instance");
    if (g->WM == 8) { if (mask) KSMI_G4CH(8, 2, 1); else if (gate) KSMI_G4CH(8, 2, 2); else KSMI_G4CH(8, 2, 0); }
    else { if (mask) KSMI_G4CH(4, 2, 1); else if (gate) KSMI_G4CH(4, 2, 2); else KSMI_G4CH(4, 2, 0); }
#undef KSMI_G4CH
  }
  if (g->deep == 1) {
#define KSMI_G4DP(WM_, NF_, EPI_)                                                                    \
  do {                                                                                               \
    auto kfn = igemm4_kernel<WM_, NF_, false, EPI_, false, 0, 2>; KSMI_IG4_DRY_RETURN; KSMI_NOTE(kfn);                                    \
    static bool attr_set = false;                                                                    \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(512), g->lds, st, ka);                                        \
    return ksmi_check_launch("igemm4");                                                              \
  } while (0)
    if (aff || g->nwv != 8 || g->NF != 2) return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm4: no deep-ring instance");
    if (g->WM == 8) { if (mask) KSMI_G4DP(8, 2, 1); else if (gate) KSMI_G4DP(8, 2, 2); else KSMI_G4DP(8, 2, 0); }
    else { if (mask) KSMI_G4DP(4, 2, 1); else if (gate) KSMI_G4DP(4, 2, 2); else KSMI_G4DP(4, 2, 0); }
#undef KSMI_G4DP
  }
  if (g->WM == 4 && g->NF == 4) KSMI_G4V(4, 4);
  if (g->WM == 4 && g->NF == 2) KSMI_G4V(4, 2);
  if (g->WM == 8 && g->NF == 4) KSMI_G4V(8, 4);
  if (g->WM == 8 && g->NF == 2) KSMI_G4V(8, 2);
#undef KSMI_G4V
#undef KSMI_G4R
#undef KSMI_G4D
#undef KSMI_G4
  return ksmi_fail(KSMI_E_UNSUPPORTED, "igemm4: no instance");
}
