// Token GEMMs, second generation (bf16): the nn.Linear layers of FloodViT (vision_transformer.py:22-31,47-50; rows = 16 x 197 tokens,
// K/N in {1024, 2048, 3072}) and the wide ChangeFormer linears, forward (NT) and input gradient (NN).
//   NT : Y[m][n]  = sum_k X[m][k]  W[n][k] + bias[n] (+ R[m][n])
//   NN : dX[m][k] = sum_n dY[m][n] W[n][k]           (+= optional)
// What changed against gemm.hip (one 32-deep K step per barrier, register-staged tiles: 320 TFLOP/s):
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4, lane-linear destination, the bank swizzle applied to the SOURCE address),
//     three stages in flight, vmcnt counted per stage -- no register staging, no ds_write;
//   * K step 64: 2 x (4 + MT) fragment reads feed 2 x 4 x MT MFMAs (v_mfma_f32_16x16x32_bf16) per barrier;
//   * the single barrier of a K step sits BETWEEN its two MFMA groups: the fragments of the next group (the second half of this
//     stage, then the first half of the next stage) are always requested before the current group's MFMAs issue;
//   * the row tile is a template parameter (TM = 32 x MT rows): 3152 token rows x N/128 column tiles fill the 256 CUs in whole
//     rounds (N = 1024: 128-row tiles, 200 WGs; N = 2048: 224 rows, 240 WGs; N = 3072: 160 rows, 480 WGs = 2 rounds).
// Swapped-operand MFMA (D = W X^T) + permuted W rows as in gemm.hip: a lane owns 16 consecutive output channels of one token row.
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"
#include "igemm_epilogue.h"      // FastDiv

namespace {

struct Gemm2P {
  const bf16_t* a; int a_rs;          // activations [rows][red]
  const bf16_t* w; int w_rs;          // weights [N][K] bf16 (row-major as nn.Linear stores them)
  const float* bias; const bf16_t* resid; int r_rs;
  bf16_t* out; int o_rs;
  int rows, red, cols, accumulate;    // reduction length, output columns
  int mtiles, ntiles;
  // ConvTranspose2d(k2, s2) as a token GEMM (round 4, models/snunet.py:32-46): a "depth row" of pixel m = (b, y, x) of the H x W map is the
  // 4C values {(dy, dx, n)} of the 2 x 2 block (2y + dy, 2x + dx) of a [B, 2H, 2W, C] NHWC tensor = two contiguous runs of 2C elements, one
  // per image row.  map_a: the A operand's rows are depth rows (input gradient: A = d out); map_o: the output rows are (forward).
  // up_C = 0: plain row-major matrices.
  int map_a, map_o, up_H, up_W, up_C, bias_mod;
};

// element offset of the depth row of pixel m, and of column c (< 4C) inside it
__device__ __forceinline__ size_t up_row_base(int m, int H, int W, int C) {
  const int x = m % W, t = m / W, y = t % H, b = t / H;
  return ((size_t)(b * 2 * H + 2 * y) * (size_t)(2 * W) + (size_t)(2 * x)) * (size_t)C;
}
__device__ __forceinline__ size_t up_col_off(int c, int W, int C) { return c < 2 * C ? (size_t)c : (size_t)(c - 2 * C) + (size_t)(2 * W) * (size_t)C; }


__device__ __forceinline__ void glds16(const unsigned char* src, unsigned dst_wave_base) {
  unsigned keep;
  dst_wave_base = __builtin_amdgcn_readfirstlane(dst_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst_wave_base) : "memory");
}
__device__ __forceinline__ void vm_wait(int n) {
#define KSMI_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    KSMI_VMW(1) KSMI_VMW(2) KSMI_VMW(3) KSMI_VMW(4) KSMI_VMW(5) KSMI_VMW(6) KSMI_VMW(7) KSMI_VMW(8) KSMI_VMW(9) KSMI_VMW(10)
    KSMI_VMW(11) KSMI_VMW(12) KSMI_VMW(13) KSMI_VMW(14) KSMI_VMW(15) KSMI_VMW(16) KSMI_VMW(17) KSMI_VMW(18) KSMI_VMW(19) KSMI_VMW(20)
    KSMI_VMW(21) KSMI_VMW(22) KSMI_VMW(23) KSMI_VMW(24) KSMI_VMW(25) KSMI_VMW(26) KSMI_VMW(27) KSMI_VMW(28) KSMI_VMW(29) KSMI_VMW(30)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef KSMI_VMW
}
// lgkmcnt(0) as a real S_WAITCNT the compiler's own wait insertion sees (an asm wait is opaque to it: it would add a second,
// later lgkmcnt(0) that also drains the fragment reads issued in between)
__device__ __forceinline__ void lgkm_wait0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
__device__ __forceinline__ void lds_barrier() {
  lgkm_wait0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 16-byte-unit swizzles of the reduction-major (transposed-read) images.  One ds_read_b64_tr_b16 has its 16 lanes read 4 rows x 4
// eight-byte pieces; rows are a multiple of 256 bytes apart (the same banks), so the 16 pieces must sit in 16 different units:
//   A side (pieces 32 bytes apart: units lu + 2 q): the row's low two bits go to unit bits 0 and 3
//   B side (pieces adjacent: 2 units): the row's low two bits go to unit bits 1 and 2
__device__ __forceinline__ int sw_a(int row) { return (row & 1) | ((row & 2) << 2); }
__device__ __forceinline__ int sw_b(int row) { return (row & 3) << 1; }
// B-side image of the weight-gradient kernel, by row length: 192-byte rows (96 columns) of 4 consecutive reduction rows already start
// in 4 different 64-byte bank groups (and a 12-unit row has no room for a 3-bit XOR), the 128- / 256-byte rows take sw_b
template <int MT> __device__ __forceinline__ int sw_bt(int row) { return MT == 3 ? 0 : sw_b(row); }

// lane's 16 consecutive outputs (acc[t][mt][r], t = 0..3) of row m -> two 16-byte stores
template <int MT>
__device__ __forceinline__ void store_row2(const Gemm2P& p, const f32x4 (&acc)[4][MT], int mt, int m, int c0, const float* bias16) {
  if (m >= p.rows) return;
  float v[16];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][mt][r] + bias16[t * 4 + r];
  bf16_t* op = p.map_o ? p.out + up_row_base(m, p.up_H, p.up_W, p.up_C) + up_col_off(c0, p.up_W, p.up_C) : p.out + (size_t)m * p.o_rs + c0;
  if (p.resid) {
    float a[8], b[8];
    vec_unpack<bf16_t>(*(const u32x4*)(p.resid + (size_t)m * p.r_rs + c0), a);
    vec_unpack<bf16_t>(*(const u32x4*)(p.resid + (size_t)m * p.r_rs + c0 + 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] += a[j]; v[8 + j] += b[j]; }
  }
  if (p.accumulate) {
    float a[8], b[8];
    vec_unpack<bf16_t>(*(const u32x4*)op, a);
    vec_unpack<bf16_t>(*(const u32x4*)(op + 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] += a[j]; v[8 + j] += b[j]; }
  }
  *(u32x4*)op = vec_pack<bf16_t>(v);
  *(u32x4*)(op + 8) = vec_pack<bf16_t>(v + 8);
}

// MT: 16-row MFMA tiles per wave along the token axis (WG = 2 x 2 waves: 32*MT token rows x 128 output columns)
// TRW: the W operand is stored reduction-major (input gradient: W[n][k], n = reduction) and read transposed (ds_read_b64_tr_b16)
template <int MT, bool TRW, int NS>
__global__ __launch_bounds__(256, 1) void gemm2_kernel(const Gemm2P p) {
  constexpr int TM = 32 * MT, KS = 64, XB = TM * 128, WB = 128 * 128, STAGE = XB + WB;
  constexpr int NQ = STAGE / 1024, NI = NQ / 4;              // DMA instructions per stage (1 KiB each) / per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int wn = wave >> 1, wm = wave & 1;
  // XCD-aware tile order: the tiles of one XCD (consecutive logical ids) walk the column tiles of a few row tiles
  const unsigned t = xcd_remap(blockIdx.x, gridDim.x);
  const int mt_i = t / p.ntiles, nt_i = t - mt_i * p.ntiles;
  const int m0 = mt_i * TM, n0 = nt_i * 128;
  const int nsteps = p.red / KS;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // ---- DMA sources: instruction q = wave + 4 i covers image rows q*8 .. q*8+7 (X rows first, then the W image)
  const unsigned char* src[NI];
  int step_bytes[NI];
  size_t jump[NI];
  const int jump_step = p.map_a ? (2 * p.up_C) / KS : 0x7fffffff;      // first K step of the second run
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = wave + 4 * i;
    jump[i] = 0;
    if (q * 8 < TM) {
      const int row = q * 8 + (lane >> 3), slot = lane & 7;
      const int chunk = slot ^ ((row >> 1) & 7);
      int m = m0 + row; if (m >= p.rows) m = p.rows - 1;
      src[i] = (const unsigned char*)(p.a + (p.map_a ? up_row_base(m, p.up_H, p.up_W, p.up_C) : (size_t)m * p.a_rs) + chunk * 8);
      step_bytes[i] = KS * 2;
      jump[i] = p.map_a ? ((size_t)(2 * p.up_W) * p.up_C - (size_t)(2 * p.up_C)) * 2 : 0;     // (second image row of the 2 x 2 block)
    } else if constexpr (!TRW) {
      const int j = q * 8 - TM + (lane >> 3), slot = lane & 7;   // W image row j holds channel perm(j): see the fragment mapping below
      const int chunk = slot ^ ((j >> 1) & 7);
      const int h = j >> 6, jj = j & 63;
      int n = n0 + h * 64 + ((jj >> 2) & 3) * 16 + (jj >> 4) * 4 + (jj & 3); if (n >= p.cols) n = p.cols - 1;
      src[i] = (const unsigned char*)(p.w + (size_t)n * p.w_rs + chunk * 8);
      step_bytes[i] = KS * 2;
    } else {
      // W image: 64 reduction rows x 256 bytes (128 output columns); 16-byte unit u of row wr holds logical unit u ^ sw_a(wr)
      const int u4 = (q * 8 - TM) * 8 + lane;                    // 16-byte unit index within the W image
      const int wr = u4 >> 4, u = u4 & 15;
      const int cb = (u ^ sw_a(wr)) << 4;
      int k = n0 + (cb >> 1); if (k + 8 > p.cols) k = p.cols - 8;
      src[i] = (const unsigned char*)(p.w + (size_t)wr * p.w_rs + k);
      step_bytes[i] = KS * p.w_rs * 2;
    }
  }
  auto issue = [&](int s, int buf) {
    const unsigned base = lds0 + (unsigned)(buf * STAGE);
#pragma unroll
    for (int i = 0; i < NI; ++i) glds16(src[i] + (size_t)s * step_bytes[i] + (s >= jump_step ? jump[i] : 0), base + (unsigned)((wave + 4 * i) * 1024));
  };

  // ---- fragment addresses (bytes within a stage) for k-substep 0; substep 1 = chunk + 4
  int fx_off[MT], fx_sw[MT], fw_off[4], fw_sw[4];
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int r = wm * 16 * MT + b * 16 + l15;
    fx_off[b] = r * 128; fx_sw[b] = (r >> 1) & 7;
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = wn * 64 + a * 16 + l15;
    fw_off[a] = XB + r * 128; fw_sw[a] = (r >> 1) & 7;
  }
  auto load_frags = [&](int buf, int sub, u32x4 (&fw)[4], u32x4 (&fx)[MT]) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int b = 0; b < MT; ++b) fx[b] = *(const u32x4*)(st + fx_off[b] + (((sub * 4 + g) ^ fx_sw[b]) << 4));
    if constexpr (!TRW) {
#pragma unroll
      for (int a = 0; a < 4; ++a) fw[a] = *(const u32x4*)(st + fw_off[a] + (((sub * 4 + g) ^ fw_sw[a]) << 4));
    } else {
      // W^T fragment of output-column tile a: lane i = l15 is MFMA row i <-> output column wn*64 + (i>>2)*16 + a*4 + (i&3);
      // its 8 reduction indices are rows sub*32 + g*8 + {jr, jr+4} .. of the image, read as two transposed 4-row pieces
      const unsigned bw = (unsigned)(uintptr_t)(st + XB);
      const int jr = l15 >> 2, q4 = l15 & 3;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int cb = (wn * 64 + q4 * 16 + a * 4) * 2;
        const int r0 = sub * 32 + g * 8 + jr, r1 = r0 + 4;
        const unsigned a0 = bw + r0 * 256 + (((cb >> 4) ^ sw_a(r0)) << 4) + (cb & 15);
        const unsigned a1 = bw + r1 * 256 + (((cb >> 4) ^ sw_a(r1)) << 4) + (cb & 15);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)a1);
        fw[a][0] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
        fw[a][1] = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
        fw[a][2] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
        fw[a][3] = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      }
    }
  };

  // two accumulator sets, one per k-substep of a stage: with both MFMA groups of an iteration chained through ONE set the
  // register allocator (ROCm 7.2) rotates every accumulator through a scratch quad (4 v_accvgpr_mov + s_nops per MFMA);
  // the sets are summed in the epilogue
  f32x4 acc[4][MT], acc2[4][MT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) { acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  auto mma_all = [&](f32x4 (&ac)[4][MT], const u32x4 (&fw)[4], const u32x4 (&fx)[MT]) {
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int a = 0; a < 4; ++a) mma16<bf16_t>(ac[a][b], fw[a], fx[b]);     // rows = channels, cols = token rows
  };

  const int pre = nsteps < NS ? nsteps : NS;
  for (int s = 0; s < pre; ++s) issue(s, s);
  vm_wait((pre - 1) * NI);
  lds_barrier();
  u32x4 fwA[4], fxA[MT], fwB[4], fxB[MT];
  load_frags(0, 0, fwA, fxA);
  int cur = 0;
  for (int s = 0; s + 1 < nsteps; ++s) {
    lgkm_wait0();                                          // set A has landed (requested one MFMA group ago)
    __builtin_amdgcn_sched_barrier(0);
    load_frags(cur, 1, fwB, fxB);
    __builtin_amdgcn_sched_barrier(0);
    mma_all(acc, fwA, fxA);
    __builtin_amdgcn_sched_barrier(0);
    // stage s+1 has landed once at most the DMAs of stages s+2 .. s+NS-1 are outstanding; after the barrier every wave holds its
    // fragments of stage s in registers, so that buffer is free for stage s+NS
    if (s + NS <= nsteps) vm_wait((NS - 2) * NI); else vm_wait((nsteps - s - 2) * NI);
    lds_barrier();
    if (s + NS < nsteps) issue(s + NS, cur);
    cur = cur + 1 == NS ? 0 : cur + 1;
    load_frags(cur, 0, fwA, fxA);
    __builtin_amdgcn_sched_barrier(0);
    mma_all(acc2, fwB, fxB);
    __builtin_amdgcn_sched_barrier(0);
  }
  load_frags(cur, 1, fwB, fxB);
  mma_all(acc, fwA, fxA);
  mma_all(acc2, fwB, fxB);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] += acc2[a][b];

  const int c0 = n0 + wn * 64 + g * 16;
  if (c0 >= p.cols) return;
  float bias16[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bias16[j] = p.bias ? p.bias[p.bias_mod ? (c0 + j) % p.bias_mod : c0 + j] : 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) store_row2<MT>(p, acc, mt, m0 + wm * 16 * MT + mt * 16 + l15, c0, bias16);
}

template <int MT, bool TRW, int NS>
int launch2n(const Gemm2P& p, hipStream_t st) {
  constexpr int lds = NS * (32 * MT * 128 + 128 * 128);
  static_assert(lds <= 160 * 1024, "LDS ring");
  auto kfn = gemm2_kernel<MT, TRW, NS>; KSMI_NOTE(kfn);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  hipLaunchKernelGGL(kfn, dim3(p.mtiles * p.ntiles), dim3(256), lds, st, p);
  return ksmi_check_launch(TRW ? "gemm2_nn" : "gemm2_nt");
}

// Tile height x ring depth, chosen from IN-SITU durations (FloodViT step under rocprofv3, tools/gemm_insitu.sh ->
// profiles/r03_gemm_insitu.txt; back-to-back launches on L2-resident operands, profiles/r03_gemm2_sweep*.txt, rank the choices
// differently: there a deeper ring buys nothing, in the step -- weights cold in HBM -- a workgroup that is alone on its CU needs the
// third stage).  One workgroup is a lock-step of DMA wait -> barrier -> fragment reads -> MFMA; a co-resident one fills the gaps.
// Registers (140 / 184 / 248 for MT 2 / 3 / 4, 312+ above) and LDS decide who can share a CU:
//   MT 2: three stages = 72 KB, two per CU, or two stages = 48 KB, three per CU;  MT 3 / 4: two stages = 56 / 64 KB, two per CU;
//   MT 4 with three stages and MT >= 5: alone.
// Cost = whole rounds of the 256 x occupancy slots x the measured round time in us at K = 1024 (only the ratios matter).
struct Tile2 { int mt, ns; };
inline Tile2 pick_tile(int rows, int ntiles) {
  static const int f_mt = ksmi_knob_int("KSMI_GEMM2_MT", 0);     // probes / tests: pin the instance
  static const int f_ns = ksmi_knob_int("KSMI_GEMM2_NS", 0);
  static const struct { int mt, ns, slots; double round_us; } cand[] = {
      {2, 3, 512, 13.0}, {4, 3, 256, 13.8}, {2, 2, 768, 15.5}, {3, 2, 512, 16.0}, {5, 3, 256, 16.3}, {6, 3, 256, 18.0},
      {4, 2, 512, 19.0}, {7, 3, 256, 21.0}, {8, 3, 256, 23.0}};
  Tile2 best = {4, 3}; double bc = 1e30;
  for (const auto& c : cand) {
    // a pinned depth of 2 or 3 selects among the candidates; the deeper, probe-only rings are applied to whatever tile wins
    if ((f_mt && c.mt != f_mt) || (f_ns && f_ns <= 3 && c.ns != f_ns)) continue;
    const long wgs = (long)((rows + 32 * c.mt - 1) / (32 * c.mt)) * ntiles;
    const double cost = (double)((wgs + c.slots - 1) / c.slots) * c.round_us;
    if (cost < bc) { bc = cost; best = {c.mt, c.ns}; }
  }
  if (f_mt && bc == 1e30) best = {f_mt, f_mt >= 5 ? 3 : 2};      // a pinned pair that is not a candidate (tests: every instance)
  if (f_ns) best.ns = f_ns;
  return best;
}

template <int MT, bool TRW>
int launch2(const Gemm2P& p, int ns, hipStream_t st) {
  constexpr int stage = 32 * MT * 128 + 128 * 128;
  if (ns <= 2) return launch2n<MT, TRW, 2>(p, st);
  if constexpr (5 * stage <= 160 * 1024) { if (ns >= 5) return launch2n<MT, TRW, 5>(p, st); }
  if constexpr (4 * stage <= 160 * 1024) { if (ns >= 4) return launch2n<MT, TRW, 4>(p, st); }
  return launch2n<MT, TRW, 3>(p, st);
}

template <bool TRW>
int dispatch2(Gemm2P& p, hipStream_t st) {
  p.ntiles = p.cols / 128;
  const Tile2 t = pick_tile(p.rows, p.ntiles);
  p.mtiles = (p.rows + 32 * t.mt - 1) / (32 * t.mt);
  switch (t.mt) {
    case 2: return launch2<2, TRW>(p, t.ns, st);
    case 3: return launch2<3, TRW>(p, t.ns, st);
    case 4: return launch2<4, TRW>(p, t.ns, st);
    case 5: return launch2<5, TRW>(p, t.ns, st);
    case 6: return launch2<6, TRW>(p, t.ns, st);
    case 7: return launch2<7, TRW>(p, t.ns, st);
    default: return launch2<8, TRW>(p, t.ns, st);
  }
}

// ---------------------------------------------------------------------------------------------- weight gradient (TN)
// out[j][i] (+)= sum_m B[m][j] A[m][i]: both operands are reduction-major ([m][columns]) and read transposed.  The A side (128
// columns per workgroup, 16 consecutive columns per lane) is the contiguous axis of the output, the B side (32*MT columns) its rows:
//   slab mode   (split reduction): A = dY (n), B = X (k):  partial[split][k][Npad + n]  -> tn_reduce_kernel (igemm.hip)
//   direct mode (one split)      : A = X (k),  B = dY (n): grad[n][k] (+)=               (row-major nn.Linear gradient)
struct Gemm2T {
  const bf16_t* a; int a_rs, a_cols;   // A-side matrix [rows][a_rs], valid columns
  const bf16_t* b; int b_rs, b_cols;
  float* out; int64_t o_rs; int64_t split_stride; int accumulate;
  int rows, rows_per_split, atiles, btiles;
  const unsigned char* zero;
  float* bias; int bias_acc;           // direct mode: bias[b column] (+)= sum over the rows of B (the nn.Linear bias gradient)
  int map_a, up_H, up_W, up_C;         // A-side rows are the depth rows of a ConvTranspose2d(k2, s2) output gradient (Gemm2P.map_a)
  float* bias_rows; int64_t bias_rs;   // slab mode (BIASA): bias_rows[split][a column] = sum over the split's rows of A (= dY): the partial
                                       // column sums of the nn.Linear bias gradient, summed over the splits by the caller
};
__device__ __attribute__((aligned(64))) unsigned char gemm2_zero_page[64];
KSMI_DEVICE_SYMBOL_GETTER(gemm2_zero, gemm2_zero_page)

// SPL = 2 (round 4): the workgroup has two 4-wave groups, each with its own LDS ring, walking one HALF of the workgroup's reduction
// range; group 1 hands its accumulators to group 0 through LDS at the end (fixed order: deterministic).  A workgroup that is alone on
// its CU (<= 256 tiles) is a lock-step of DMA wait -> barrier -> fragment reads -> MFMA; the second group doubles the waves per SIMD
// and halves the serial K loop without partial slabs in memory and without a reducer launch.
// BIASA (round 5): slab mode also emits the column sums of the A side (dY) -- the bias gradient of the layer -- per split: one MFMA per
// A fragment against an all-ones B fragment in the workgroups of B tile 0 (waves of B-side group 0), instead of a channel_sum pass that
// re-reads dY from HBM (ChangeFormer: 40 launches, ~0.5 ms per step).  A template parameter: the direct-mode instances of the ViT step
// compile as before.
template <int MT, int NS, int SPL, bool BIASA = false>
__global__ __launch_bounds__(256 * SPL, 1) void gemm2_tn_kernel(const Gemm2T p) {
  static_assert(!BIASA || SPL == 1, "A-side column sums: one wave group");
  constexpr int TB = 32 * MT, KS = 64, AB = 64 * 256, BROW = TB * 2, BB = 64 * BROW, STAGE = AB + BB;
  constexpr int NQ = STAGE / 1024, NI = NQ / 4, BGM = BROW / 32 - 1, AU = 16, BU = BROW / 16;
  static_assert(NQ % 4 == 0, "whole DMA rounds");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int grp = SPL == 1 ? 0 : wave_all >> 2, wave = wave_all & 3;
  const int wn = wave >> 1, wm = wave & 1;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int bt = tile / p.atiles, at = tile - bt * p.atiles;
  const int a0 = at * 128, b0 = bt * TB;
  const int wg_begin = split * p.rows_per_split;
  const int wg_end = min(p.rows, wg_begin + p.rows_per_split);
  const int steps_wg = (max(wg_end - wg_begin, 0) + KS - 1) / KS;
  const int nsteps = (steps_wg + SPL - 1) / SPL;                  // the same count in every group (common barriers); rows past the group's end read zeros
  const int m_begin = wg_begin + grp * nsteps * KS;
  const int m_end = min(wg_end, m_begin + nsteps * KS);
  const unsigned lds0 = (unsigned)(uintptr_t)smem + (unsigned)(grp * NS * STAGE);

  // DMA: instruction q = wave + 4 i; 16-byte unit index within the stage -> (image, reduction row, column granule)
  const unsigned char* src[NI];
  int srow[NI], sstep[NI];
  bool amap[NI];
  const FastDiv dW(p.up_W > 0 ? p.up_W : 1), dH(p.up_H > 0 ? p.up_H : 1);       // (exact: rows * W < 2^32)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = wave + 4 * i;
    amap[i] = false;
    if (q * 1024 < AB) {
      const int u4 = q * 64 + lane, wr = u4 / AU, u = u4 % AU;
      const int cb = (u ^ sw_a(wr)) << 4;
      int c = a0 + (cb >> 1); if (c + 8 > p.a_cols) c = p.a_cols - 8;
      src[i] = p.map_a ? (const unsigned char*)(p.a + up_col_off(c, p.up_W, p.up_C))        // (+ the row base, per step: issue())
                       : (const unsigned char*)(p.a + (size_t)(m_begin + wr) * p.a_rs + c);
      srow[i] = wr; sstep[i] = KS * p.a_rs * 2; amap[i] = p.map_a != 0;
    } else {
      const int u4 = q * 64 - AB / 16 + lane, wr = u4 / BU, u = u4 % BU;
      const int cb = (u ^ sw_bt<MT>(wr)) << 4;
      int c = b0 + (cb >> 1); if (c + 8 > p.b_cols) c = p.b_cols - 8;
      src[i] = (const unsigned char*)(p.b + (size_t)(m_begin + wr) * p.b_rs + c);
      srow[i] = wr; sstep[i] = KS * p.b_rs * 2;
    }
  }
  auto issue = [&](int s, int buf) {
    const unsigned base = lds0 + (unsigned)(buf * STAGE);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // reduction rows past the split's end read the zero page (two 32-bit selects: a pointer select compiles to two masked DMAs)
      const bool ok = m_begin + s * KS + srow[i] < m_end;
      uint64_t real = (uint64_t)(uintptr_t)(src[i] + (size_t)s * sstep[i]);
      if (amap[i]) {                    // depth row of pixel m = (b, y, x): two integer divisions by magic numbers per piece
        const int m = ok ? m_begin + s * KS + srow[i] : 0;
        const int t = dW.div(m), x = m - t * p.up_W, b = dH.div(t), y = t - b * p.up_H;
        real = (uint64_t)(uintptr_t)(src[i] + (((size_t)(b * 2 * p.up_H + 2 * y) * (size_t)(2 * p.up_W) + (size_t)(2 * x)) * (size_t)p.up_C) * 2);
      }
      const uint64_t zp = (uint64_t)(uintptr_t)p.zero;
      const uint32_t lo = ok ? (uint32_t)real : (uint32_t)zp, hi = ok ? (uint32_t)(real >> 32) : (uint32_t)(zp >> 32);
      glds16((const unsigned char*)(uintptr_t)(((uint64_t)hi << 32) | lo), base + (unsigned)((wave + 4 * i) * 1024));
    }
  };
  auto tr8 = [&](unsigned img, int row_bytes, int sw, int r0, int cb) -> u32x4 {      // sw = swizzle of rows r0 and r0 + 4 (same low bits)
    const int r1 = r0 + 4;
    const unsigned x0 = img + r0 * row_bytes + (((cb >> 4) ^ sw) << 4) + (cb & 15);
    const unsigned x1 = img + r1 * row_bytes + (((cb >> 4) ^ sw) << 4) + (cb & 15);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)x0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)x1);
    u32x4 f;
    f[0] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    f[1] = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    f[2] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    f[3] = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return f;
  };
  auto load_frags = [&](int buf, int sub, u32x4 (&fa)[4], u32x4 (&fb)[MT]) {
    const unsigned st = lds0 + (unsigned)(buf * STAGE);
    const int jr = l15 >> 2, q4 = l15 & 3, r0 = sub * 32 + g * 8 + jr;
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = tr8(st, 256, sw_a(r0), r0, (wn * 64 + q4 * 16 + a * 4) * 2);         // lane i <-> column wn*64 + (i>>2)*16 + a*4 + (i&3)
#pragma unroll
    for (int b = 0; b < MT; ++b) fb[b] = tr8(st + AB, BROW, sw_bt<MT>(r0), r0, (wm * 16 * MT + b * 16 + q4 * 4) * 2);   // lane i <-> column wm*16*MT + b*16 + i
  };
  f32x4 acc[4][MT], acc2[4][MT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) { acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  // bias gradient = column sums of the B side: one extra MFMA per B fragment against an all-ones A fragment, in the workgroups of
  // A tile 0 and the waves of A-side group 0 only (the other waves hold the same B fragments)
  const bool do_bias = p.bias != nullptr && at == 0 && wn == 0;
  f32x4 accb[MT], accb2[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) { accb[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; accb2[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bool do_biasA = BIASA && p.bias_rows != nullptr && bt == 0 && wm == 0;
  f32x4 accA[BIASA ? 4 : 1], accA2[BIASA ? 4 : 1];
#pragma unroll
  for (int a = 0; a < (BIASA ? 4 : 1); ++a) { accA[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; accA2[a] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  auto mma_all = [&](f32x4 (&ac)[4][MT], f32x4 (&ab)[MT], f32x4 (&aa)[BIASA ? 4 : 1], const u32x4 (&fa)[4], const u32x4 (&fb)[MT]) {
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int a = 0; a < 4; ++a) mma16<bf16_t>(ac[a][b], fa[a], fb[b]);
    if (do_bias) {
#pragma unroll
      for (int b = 0; b < MT; ++b) mma16<bf16_t>(ab[b], ones, fb[b]);
    }
    if constexpr (BIASA) {
      if (do_biasA) {
#pragma unroll
        for (int a = 0; a < 4; ++a) mma16<bf16_t>(aa[a], fa[a], ones);         // D[i][j] = sum_k A[i][k]: the same in every column j
      }
    }
  };
  if (nsteps > 0) {
    const int pre = nsteps < NS ? nsteps : NS;
    for (int s = 0; s < pre; ++s) issue(s, s);
    vm_wait((pre - 1) * NI);
    lds_barrier();
    u32x4 faA[4], fbA[MT], faB[4], fbB[MT];
    load_frags(0, 0, faA, fbA);
    int cur = 0;
    for (int s = 0; s + 1 < nsteps; ++s) {
      lgkm_wait0();
      __builtin_amdgcn_sched_barrier(0);
      load_frags(cur, 1, faB, fbB);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(acc, accb, accA, faA, fbA);
      __builtin_amdgcn_sched_barrier(0);
      if (s + NS <= nsteps) vm_wait((NS - 2) * NI); else vm_wait((nsteps - s - 2) * NI);
      lds_barrier();
      if (s + NS < nsteps) issue(s + NS, cur);
      cur = cur + 1 == NS ? 0 : cur + 1;
      load_frags(cur, 0, faA, fbA);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(acc2, accb2, accA2, faB, fbB);
      __builtin_amdgcn_sched_barrier(0);
    }
    load_frags(cur, 1, faB, fbB);
    mma_all(acc, accb, accA, faA, fbA);
    mma_all(acc2, accb2, accA2, faB, fbB);
  }
  if constexpr (SPL == 2) {
    // group 1 -> group 0: [(a * MT + b)][wave * 64 + lane] float4 (+ the bias rows behind them), in the ring memory
    __syncthreads();
    f32x4* xch = (f32x4*)smem;
    const int slot = wave * 64 + lane;
    if (grp == 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) xch[(a * MT + b) * 256 + slot] = acc[a][b] + acc2[a][b];
#pragma unroll
      for (int b = 0; b < MT; ++b) xch[(4 * MT + b) * 256 + slot] = accb[b] + accb2[b];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) { acc[a][b] = (acc[a][b] + acc2[a][b]) + xch[(a * MT + b) * 256 + slot]; acc2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int b = 0; b < MT; ++b) { accb[b] = (accb[b] + accb2[b]) + xch[(4 * MT + b) * 256 + slot]; accb2[b] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  }
  if constexpr (BIASA) {
    // lane (g, l15 = 0): rows g*4 + r of fragment a <-> A-side columns a0 + wn*64 + g*16 + a*4 + r: 16 consecutive floats
    if (do_biasA && l15 == 0) {
      float* br = p.bias_rows + (int64_t)split * p.bias_rs;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4 v = accA[a] + accA2[a];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = a0 + wn * 64 + g * 16 + a * 4 + r;
          if (c < p.a_cols) br[c] = v[r];
        }
      }
    }
  }
  if (do_bias && g == 0) {                                // every accumulator row holds the column sum: take row 0 (g = 0, r = 0)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      const int bc = b0 + wm * 16 * MT + b * 16 + l15;
      if (bc < p.b_cols) { const float v = accb[b][0] + accb2[b][0]; p.bias[bc] = p.bias_acc ? p.bias[bc] + v : v; }
    }
  }
  // lane: 16 consecutive A-side columns (acc[a][.][r]: column g*16 + a*4 + r) of B-side column wm*16*MT + b*16 + l15
  const int ac0 = a0 + wn * 64 + g * 16;
  if (ac0 >= p.a_cols) return;
  float* ob = p.out + (size_t)split * p.split_stride;
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int bc = b0 + wm * 16 * MT + b * 16 + l15;
    if (bc >= p.b_cols) continue;
    float* o = ob + (int64_t)bc * p.o_rs + ac0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4 v = acc[a][b] + acc2[a][b];
      if (ac0 + a * 4 + 4 <= p.a_cols) {
        if (p.accumulate) { const f32x4 old = *(const f32x4*)(o + a * 4); v += old; }
        *(f32x4*)(o + a * 4) = v;
      } else {
        for (int r = 0; r < 4; ++r)
          if (ac0 + a * 4 + r < p.a_cols) o[a * 4 + r] = p.accumulate ? o[a * 4 + r] + v[r] : v[r];
      }
    }
  }
}

}  // namespace

// 0 = launched, 1 = shape not covered (the caller falls back to gemm.hip), < 0 = error
int ksmi_gemm2_nt(const void* x, int x_rs, const void* w, int w_rs, const float* bias, const void* resid, int r_rs, void* y, int y_rs,
                  int rows, int K, int N, hipStream_t st) {
  static const bool off = ksmi_knob_is_set("KSMI_GEMM2_OFF");
  if (off || K % 64 || N % 128 || rows < 64 || K < 128) return 1;
  Gemm2P p = {(const bf16_t*)x, x_rs, (const bf16_t*)w, w_rs, bias, (const bf16_t*)resid, r_rs, (bf16_t*)y, y_rs, rows, K, N, 0, 0, 0};
  return dispatch2<false>(p, st);
}

int ksmi_gemm2_nn(const void* dy, int dy_rs, const void* w, int w_rs, void* dx, int dx_rs, int rows, int K, int N, int accumulate,
                  hipStream_t st) {
  static const bool off = ksmi_knob_is_set("KSMI_GEMM2_OFF");
  if (off || N % 64 || K % 128 || rows < 64 || N < 128) return 1;
  Gemm2P p = {(const bf16_t*)dy, dy_rs, (const bf16_t*)w, w_rs, nullptr, nullptr, 0, (bf16_t*)dx, dx_rs, rows, N, K, accumulate, 0, 0};
  return dispatch2<true>(p, st);
}

// ConvTranspose2d(k2, s2) forward / input gradient as token GEMMs over depth rows (Gemm2P.map_*): wb = [4C][C] bf16, row (d, n) = Wt[:, n, d]
int ksmi_gemm2_up_forward(const void* x, const void* wb, const float* bias, void* y, int B, int H, int W, int C, hipStream_t st) {
  if (C % 64 || C < 128 || B * H * W < 64) return 1;
  Gemm2P p = {(const bf16_t*)x, C, (const bf16_t*)wb, C, bias, nullptr, 0, (bf16_t*)y, 0, B * H * W, C, 4 * C, 0, 0, 0};
  p.map_o = 1; p.up_H = H; p.up_W = W; p.up_C = C; p.bias_mod = C;
  return dispatch2<false>(p, st);
}
int ksmi_gemm2_up_dgrad(const void* dy, const void* wb, void* dx, int accumulate, int B, int H, int W, int C, hipStream_t st) {
  if (C % 128 || C < 128 || B * H * W < 64) return 1;
  Gemm2P p = {(const bf16_t*)dy, 0, (const bf16_t*)wb, C, nullptr, nullptr, 0, (bf16_t*)dx, C, B * H * W, 4 * C, C, accumulate, 0, 0};
  p.map_a = 1; p.up_H = H; p.up_W = W; p.up_C = C;
  return dispatch2<true>(p, st);
}

template <int MT, int NS, int SPL = 1, bool BIASA = false>
static void launch_tn(dim3 grid, const Gemm2T& p, hipStream_t st) {
  constexpr int lds = SPL * NS * (64 * 256 + 64 * 64 * MT);
  static_assert(lds <= 160 * 1024, "LDS ring");
  static_assert(SPL == 1 || (4 * MT + MT) * 256 * 16 <= lds, "exchange area");
  auto kfn = gemm2_tn_kernel<MT, NS, SPL, BIASA>; KSMI_NOTE(kfn);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  hipLaunchKernelGGL(kfn, grid, dim3(256 * SPL), lds, st, p);
}

// two wave groups per workgroup (gemm2_tn_kernel SPL = 2) pay when a workgroup would be alone on its CU and its K loop is long enough
bool ksmi_gemm2_tn_spl(int tiles_times_splits, int steps_per_split) {
  // KSMI_TN_SPL = 1: wherever it fits (instance tests), 2: by shape, unset / 0: off.  OFF by default: measured in the FloodViT step
  // (profiles/r04_tn_spl.txt) the two-group kernels are 11-14 % shorter (38.9 -> 33.4 us, 61.8 -> 54.7 us) and the step is 3.6 % SLOWER
  // (13.81 -> 14.31 ms): a 512-thread workgroup with 144 KB of LDS owns its CU, while the one-group instance (72 KB) shares it with a
  // token GEMM of the main stream -- on the side stream co-residency is worth more than the shorter kernel (MAE: +0.9 %).
  static const int f = ksmi_knob_int("KSMI_TN_SPL", 0);
  if (f <= 0) return false;
  if (f == 1) return steps_per_split >= 2;
  return tiles_times_splits <= 256 && steps_per_split >= 12;
}

bool ksmi_gemm2_tn_enabled(int K, int N, int rows_per_split) {
  static const bool off = ksmi_knob_is_set("KSMI_GEMM2_OFF") || ksmi_knob_is_set("KSMI_GEMM2_TN_OFF");
  return !(off || K % 8 || N % 8 || K < 64 || N < 64 || rows_per_split % 64);
}

// weight gradient of a plain nn.Linear: x [rows][K] (row stride x_rs), dy [rows][N]; `slab` = split partial sums [nsplit][K][Npad]
// (n contiguous: the layout of tn_reduce_kernel), or nsplit = 1 and grad [N][g_rs] written directly.  0 launched / 1 not covered
int ksmi_gemm2_tn(const void* x, int x_rs, const void* dy, int dy_rs, float* slab, int npad, float* grad, int64_t g_rs, int rows, int K, int N,
                  int Kslab, int nsplit, int rows_per_split, int btile, int accumulate, float* bias_grad, int bias_accumulate, hipStream_t st) {
  if (!ksmi_gemm2_tn_enabled(K, N, rows_per_split)) return 1;
  const unsigned char* const zero_page = gemm2_zero();
  if (!zero_page) return ksmi_fail(KSMI_E_UNSUPPORTED, "gemm2: zero page");
  Gemm2T p = {};
  p.rows = rows; p.rows_per_split = rows_per_split; p.zero = zero_page;
  if (nsplit == 1 && grad) {        // direct: A = x (k contiguous in grad), B = dy
    p.a = (const bf16_t*)x; p.a_rs = x_rs; p.a_cols = K; p.b = (const bf16_t*)dy; p.b_rs = dy_rs; p.b_cols = N;
    p.out = grad; p.o_rs = g_rs; p.split_stride = 0; p.accumulate = accumulate;
    p.bias = bias_grad; p.bias_acc = bias_accumulate;
  } else {                          // slabs: A = dy (n contiguous in the slab), B = x
    p.a = (const bf16_t*)dy; p.a_rs = dy_rs; p.a_cols = N; p.b = (const bf16_t*)x; p.b_rs = x_rs; p.b_cols = K;
    p.out = slab; p.o_rs = npad; p.split_stride = (int64_t)Kslab * npad; p.accumulate = 0;
    p.bias_rows = bias_grad; p.bias_rs = N;      // (slab mode: partial column sums of dY per split, [nsplit][N]: ksmi_conv_wgrad_fuses_bias == 2)
  }
  p.atiles = (p.a_cols + 127) / 128; p.btiles = (p.b_cols + btile - 1) / btile;
  static const int ns = ksmi_knob_int("KSMI_TN_NS", 3);      // probes: ring depth of the weight-gradient kernel
  const dim3 grid(p.atiles * p.btiles, nsplit);
  if (!p.bias_rows && ns <= 3 && ksmi_gemm2_tn_spl((int)(grid.x * grid.y), (rows_per_split + 63) / 64)) {
    if (btile == 128) launch_tn<4, 2, 2>(grid, p, st);
    else if (btile == 96) launch_tn<3, 2, 2>(grid, p, st);
    else if (btile == 64) launch_tn<2, 3, 2>(grid, p, st);
    else return ksmi_fail(KSMI_E_ARG, "gemm2_tn: B-side tile must be 64, 96 or 128");
    return ksmi_check_launch("gemm2_tn");
  }
  if (p.bias_rows) {                // A-side column sums: the three-stage one-group instances
    if (btile == 128) launch_tn<4, 3, 1, true>(grid, p, st);
    else if (btile == 96) launch_tn<3, 3, 1, true>(grid, p, st);
    else if (btile == 64) launch_tn<2, 3, 1, true>(grid, p, st);
    else return ksmi_fail(KSMI_E_ARG, "gemm2_tn: B-side tile must be 64, 96 or 128");
    return ksmi_check_launch("gemm2_tn");
  }
  if (btile == 128) { if (ns >= 5) launch_tn<4, 5>(grid, p, st); else if (ns == 4) launch_tn<4, 4>(grid, p, st); else launch_tn<4, 3>(grid, p, st); }
  else if (btile == 96) { if (ns >= 5) launch_tn<3, 5>(grid, p, st); else if (ns == 4) launch_tn<3, 4>(grid, p, st); else launch_tn<3, 3>(grid, p, st); }
  else if (btile == 64) { if (ns >= 6) launch_tn<2, 6>(grid, p, st); else if (ns == 5) launch_tn<2, 5>(grid, p, st); else if (ns == 4) launch_tn<2, 4>(grid, p, st); else launch_tn<2, 3>(grid, p, st); }
  else return ksmi_fail(KSMI_E_ARG, "gemm2_tn: B-side tile must be 64, 96 or 128");
  return ksmi_check_launch("gemm2_tn");
}


// ---------------------------------------------------------------------------------------------- ConvTranspose2d(k2, s2) as token GEMMs
namespace {
// wb[(d * C + n) * C + c] = bf16(Wt[c][n][d])   (nn.ConvTranspose2d weight [C_in = C][C_out = C][2][2], d = dy * 2 + dx)
__global__ void up_pack_kernel(const float* __restrict__ wt, bf16_t* __restrict__ wb, int C) {
  const int64_t n_el = (int64_t)4 * C * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C); const int64_t r = e / C; const int n = (int)(r % C), d = (int)(r / C);
    wb[e] = f32_to_bf16(wt[((int64_t)c * C + n) * 4 + d]);
  }
}
// every `up` weight of a plan in one launch (blockIdx.y = tensor; the six SNUNet packs were six 6-10 us launches)
struct UpPackBatch { const float* wt[KSMI_UP_PACK_MAX]; bf16_t* wb[KSMI_UP_PACK_MAX]; int C[KSMI_UP_PACK_MAX]; };
__global__ void up_pack_batched_kernel(const UpPackBatch b) {
  const float* __restrict__ wt = b.wt[blockIdx.y];
  bf16_t* __restrict__ wb = b.wb[blockIdx.y];
  const int C = b.C[blockIdx.y];
  const int64_t n_el = (int64_t)4 * C * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_el; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C); const int64_t r = e / C; const int n = (int)(r % C), d = (int)(r / C);
    wb[e] = f32_to_bf16(wt[((int64_t)c * C + n) * 4 + d]);
  }
}
// grad[c][n][d] (+)= sum_split slab[split][c][d * C + n]   (slab rows k = c, columns n' = (d, n); fixed order => deterministic)
// The gradient is small (C x C x 4) and the splits are many (C = 64: 256 slabs), so one thread per (c, n) walking every split is a
// 16-workgroup launch of 256 dependent strided reads (40-58 us, round-4 profiles).  Here a workgroup owns one input channel c and NB
// output channels: its threads are SL slab lanes x 4 depths x NQ float4 columns; slab lane sl sums the splits sl, sl + SL, ... (two
// accumulators), the lanes are combined through LDS in ascending order and written with d innermost.
template <int SL>
__global__ __launch_bounds__(256) void up_wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int C, float* __restrict__ grad, int accumulate) {
  constexpr int NQ = 64 / SL;                              // float4 columns per depth (SL * 4 * NQ = 256 threads)
  constexpr int NB = NQ * 4;                               // output channels per workgroup (16 | 64; C % 64 == 0)
  __shared__ f32x4 red[SL][4][NQ];
  const int tid = threadIdx.x;
  const int q = tid % NQ, d = (tid / NQ) & 3, sl = tid / (NQ * 4);
  const int nblk = C / NB;
  const int c = blockIdx.x / nblk, n0 = (blockIdx.x - c * nblk) * NB;
  const int64_t ss = (int64_t)C * 4 * C;
  const float* p = slab + (int64_t)c * 4 * C + d * C + n0 + 4 * q;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  int sp = sl;
  for (; sp + SL < nsplit; sp += 2 * SL) { a0 += *(const f32x4*)(p + (int64_t)sp * ss); a1 += *(const f32x4*)(p + (int64_t)(sp + SL) * ss); }
  if (sp < nsplit) a0 += *(const f32x4*)(p + (int64_t)sp * ss);
  red[sl][d][q] = a0 + a1;
  __syncthreads();
  if (tid < NB) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < SL; ++l)
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) v[dd] += red[l][dd][tid >> 2][tid & 3];
    f32x4* o = (f32x4*)(grad + ((int64_t)c * C + n0 + tid) * 4);
    *o = accumulate ? *o + v : v;
  }
}
static void launch_up_wgrad_reduce(const float* slab, int nsplit, int C, float* grad, int accumulate, hipStream_t st) {
  if (nsplit >= 16) hipLaunchKernelGGL(up_wgrad_reduce_kernel<16>, dim3((unsigned)(C * (C / 16))), dim3(256), 0, st, slab, nsplit, C, grad, accumulate);
  else hipLaunchKernelGGL(up_wgrad_reduce_kernel<4>, dim3((unsigned)(C * (C / 64))), dim3(256), 0, st, slab, nsplit, C, grad, accumulate);
}
inline void up_wgrad_geom(int rows, int C, int* nsplit, int* rps) {
  const int tiles = (4 * C / 128) * (C / 64);
  int want = (512 + tiles - 1) / tiles;                    // ~two workgroups per CU
  const int steps_all = (rows + 63) / 64;
  if (want > steps_all / 8) want = steps_all / 8 > 0 ? steps_all / 8 : 1;      // (>= 8 K steps per split)
  if (want < 1) want = 1;
  *rps = ((steps_all + want - 1) / want) * 64;
  *nsplit = (rows + *rps - 1) / *rps;
}
}  // namespace

int ksmi_gemm2_up_pack(const float* wt, void* wb, int C, hipStream_t st) {
  const int64_t n_el = (int64_t)4 * C * C;
  hipLaunchKernelGGL(up_pack_kernel, dim3((unsigned)((n_el + 255) / 256 > 2048 ? 2048 : (n_el + 255) / 256)), dim3(256), 0, st, wt, (bf16_t*)wb, C);
  return ksmi_check_launch("up_pack");
}
size_t ksmi_gemm2_up_wgrad_workspace(int B, int H, int W, int C) {
  int nsplit, rps;
  up_wgrad_geom(B * H * W, C, &nsplit, &rps);
  return (size_t)nsplit * C * 4 * C * sizeof(float);
}
// dWt[c][n][d] (+)= sum_m x[m][c] dY_depth[m][(d, n)]: slab-mode gemm2_tn (A = the depth rows of d out, B = x) + the permuting reducer
int ksmi_gemm2_up_wgrad(const void* x, const void* dy, float* slab, float* grad, int accumulate, int B, int H, int W, int C, hipStream_t st) {
  if (C % 64 || C < 64 || B * H * W < 64) return 1;          // (a 128-column A tile stays inside one 2C-element run of the depth row)
  const unsigned char* const zero_page = gemm2_zero();
  if (!zero_page) return ksmi_fail(KSMI_E_UNSUPPORTED, "gemm2: zero page");
  const int rows = B * H * W;
  int nsplit, rps;
  up_wgrad_geom(rows, C, &nsplit, &rps);
  Gemm2T p = {};
  p.rows = rows; p.rows_per_split = rps; p.zero = zero_page;
  p.a = (const bf16_t*)dy; p.a_rs = 0; p.a_cols = 4 * C; p.b = (const bf16_t*)x; p.b_rs = C; p.b_cols = C;
  p.out = slab; p.o_rs = 4 * C; p.split_stride = (int64_t)C * 4 * C; p.accumulate = 0;
  p.map_a = 1; p.up_H = H; p.up_W = W; p.up_C = C;
  p.atiles = 4 * C / 128; p.btiles = C / 64;
  launch_tn<2, 3>(dim3(p.atiles * p.btiles, nsplit), p, st);
  int rc = ksmi_check_launch("gemm2_up_wgrad");
  if (rc) return rc;
  launch_up_wgrad_reduce(slab, nsplit, C, grad, accumulate, st);
  return ksmi_check_launch("up_wgrad_reduce");
}

extern "C" {
int ksmi_up_gemm_supported(int B, int H, int W, int C, int dtype) {
  static const bool off = ksmi_knob_is_set("KSMI_GEMM2_OFF");
  return !off && dtype == KSMI_BF16 && C >= 128 && C % 128 == 0 && (int64_t)B * H * W >= 64 && (int64_t)B * H * W * 112 < ((int64_t)1 << 32) ? 1 : 0;
}
int ksmi_up_wgrad_supported(int B, int H, int W, int C, int dtype) {
  static const bool off = ksmi_knob_is_set("KSMI_GEMM2_OFF");
  return !off && dtype == KSMI_BF16 && C >= 64 && C % 64 == 0 && (int64_t)B * H * W >= 64 && (int64_t)B * H * W * 112 < ((int64_t)1 << 32) ? 1 : 0;
}
int ksmi_up_pack_weight(const float* wt, void* wb, int C, void* stream) { return ksmi_gemm2_up_pack(wt, wb, C, (hipStream_t)stream); }
int ksmi_up_pack_weights_batched(const float* const* wt, void* const* wb, const int* C, int n, void* stream) {
  if (!wt || !wb || !C || n < 1 || n > KSMI_UP_PACK_MAX) return ksmi_fail(KSMI_E_ARG, "up_pack_weights_batched: 1 .. KSMI_UP_PACK_MAX tensors");
  UpPackBatch b = {};
  for (int i = 0; i < n; ++i) {
    if (!wt[i] || !wb[i] || C[i] < 1) return ksmi_fail(KSMI_E_ARG, "up_pack_weights_batched: bad entry");
    b.wt[i] = wt[i]; b.wb[i] = (bf16_t*)wb[i]; b.C[i] = C[i];
  }
  hipLaunchKernelGGL(up_pack_batched_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, b);
  return ksmi_check_launch("up_pack_weights_batched");
}
int ksmi_up_forward(const void* x, const void* wb, const float* bias, void* y, int B, int H, int W, int C, void* stream) {
  const int rc = ksmi_gemm2_up_forward(x, wb, bias, y, B, H, W, C, (hipStream_t)stream);
  return rc == 1 ? ksmi_fail(KSMI_E_UNSUPPORTED, "up_forward: shape not covered (ksmi_up_gemm_supported)") : rc;
}
int ksmi_up_dgrad(const void* dy, const void* wb, void* dx, int accumulate, int B, int H, int W, int C, void* stream) {
  const int rc = ksmi_gemm2_up_dgrad(dy, wb, dx, accumulate, B, H, W, C, (hipStream_t)stream);
  return rc == 1 ? ksmi_fail(KSMI_E_UNSUPPORTED, "up_dgrad: shape not covered (ksmi_up_gemm_supported)") : rc;
}
size_t ksmi_up_wgrad_workspace(int B, int H, int W, int C) { return ksmi_gemm2_up_wgrad_workspace(B, H, W, C); }
int ksmi_up_wgrad(const void* x, const void* dy, float* workspace, float* grad, int accumulate, int B, int H, int W, int C, void* stream) {
  const int rc = ksmi_gemm2_up_wgrad(x, dy, workspace, grad, accumulate, B, H, W, C, (hipStream_t)stream);
  return rc == 1 ? ksmi_fail(KSMI_E_UNSUPPORTED, "up_wgrad: shape not covered (ksmi_up_gemm_supported)") : rc;
}
}  // extern "C"
