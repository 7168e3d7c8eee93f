// Device-memory rate probes behind ksmi_hbm_probe (measurement aid of bench.py: `roofline.measured_peaks`).  SURVEY.md §8(d) asks for
// the achieved fractions against the spec figure AND against what THIS part delivers: the probes stream operands far larger than
// the 256 MB memory-side cache and are timed with HIP events by the caller.
//   mode 0  read, LDS-DMA     every wave keeps 8 x 1 KB global_load_lds_dwordx4 in flight into its own LDS slice, nothing consumes
//                             them: the pure read stream of the persistent convolution kernels (dma.h)
//   mode 1  read, vector      16-byte non-temporal loads, 8 per lane in flight, xor-folded into one store per workgroup
//   mode 2  copy              b[i] = a[i]                     (bytes moved = 2 x n)
//   mode 3  triad             c[i] = a[i] + s * b[i]  (fp32)  (bytes moved = 3 x n)
//   mode 4  fill              a[i] = s                        (bytes moved = n)
#include "common.h"
#include "dma.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

constexpr int kT = 512;

// wmask != 0 (round 5, "LDS fill" probe): the byte offsets wrap inside a window of wmask + 1 bytes (a power of two), so the same
// nbytes of LDS-DMA traffic is served by the L2 (window <= a few MB) or the memory-side cache (<= 256 MB) instead of HBM: the rate at
// which a CU's LDS can be FILLED when the data is on chip -- the roof of every LDS-staged kernel that re-reads weights / halo overlap.
// shared != 0: every workgroup walks the SAME addresses (the weight slab all CUs re-read), else disjoint slices of the window.
__global__ __launch_bounds__(kT) void probe_read_dma(const unsigned char* __restrict__ a, size_t nbytes, unsigned* sink, size_t wmask, int shared) {
  extern __shared__ unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kT / 64;
  const size_t chunk = (size_t)1024 * 8;                                    // bytes per wave per trip: 8 DMA instructions of 1 KB
  const size_t per_wg = chunk * nw;
  const unsigned lds0 = (unsigned)(uintptr_t)smem + wave * 8192;
  const size_t trips = nbytes / (per_wg * gridDim.x);
  size_t off = (shared ? 0 : (size_t)blockIdx.x * per_wg) + (size_t)wave * chunk + lane * 16;
  const size_t step = shared ? per_wg : per_wg * gridDim.x;
  for (size_t t = 0; t < trips; ++t) {
    const unsigned char* p = a + (wmask ? (off & wmask) : off);
#pragma unroll
    for (int k = 0; k < 8; ++k) glds16_flat(p + k * 1024, lds0 + k * 1024);
    off += step;
    vm_wait(8);                                                             // the previous trip has landed; this one stays in flight
  }
  vm_wait(0);
  __syncthreads();
  if (threadIdx.x == 0 && smem[17] == 0x5a && smem[4097] == 0xa5) sink[0] = 1;   // (keeps the loads observable)
}

__global__ __launch_bounds__(256) void probe_read_vec(const u32x4* __restrict__ a, size_t nvec, unsigned* sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; v + 7 * stride < nvec; v += 8 * stride) {
    u32x4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = __builtin_nontemporal_load(a + v + k * stride);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= r[k];
  }
  for (; v < nvec; v += stride) acc ^= __builtin_nontemporal_load(a + v);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[1] = 1;
}

template <bool NT, typename V> __device__ __forceinline__ void st16(V* p, const V& v) {
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void probe_copy(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t nvec) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; v + 3 * stride < nvec; v += 4 * stride) {
    u32x4 r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = NT ? __builtin_nontemporal_load(a + v + k * stride) : a[v + k * stride];
#pragma unroll
    for (int k = 0; k < 4; ++k) st16<NT>(b + v + k * stride, r[k]);
  }
  for (; v < nvec; v += stride) b[v] = a[v];
}

template <bool NT>
__global__ __launch_bounds__(256) void probe_triad(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, float s, size_t nvec) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; v + stride < nvec; v += 2 * stride) {
    const f32x4 a0 = a[v], b0 = b[v], a1 = a[v + stride], b1 = b[v + stride];
    st16<NT>(c + v, a0 + s * b0);
    st16<NT>(c + v + stride, a1 + s * b1);
  }
  for (; v < nvec; v += stride) c[v] = a[v] + s * b[v];
}

template <bool NT>
__global__ __launch_bounds__(256) void probe_fill(f32x4* __restrict__ a, float s, size_t nvec) {
  const f32x4 val = {s, s, s, s};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; v + 3 * stride < nvec; v += 4 * stride) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st16<NT>(a + v + k * stride, val);
  }
  for (; v < nvec; v += stride) a[v] = val;
}

// contiguous form: workgroup b owns vectors [b * per, (b + 1) * per) and walks them front to back (what a flat elementwise launch does)
template <int KIND>
__global__ __launch_bounds__(256) void probe_chunked(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, float s, size_t nvec) {
  const size_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const size_t v0 = (size_t)blockIdx.x * per, v1 = v0 + per < nvec ? v0 + per : nvec;
  const f32x4 val = {s, s, s, s};
  size_t v = v0 + threadIdx.x;
  for (; v + 3 * 256 < v1; v += 4 * 256) {
    if (KIND == 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) c[v + k * 256] = val;
    } else if (KIND == 2) {
      f32x4 r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] = a[v + k * 256];
#pragma unroll
      for (int k = 0; k < 4; ++k) c[v + k * 256] = r[k];
    } else {
      f32x4 r[4], q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { r[k] = a[v + k * 256]; q[k] = b[v + k * 256]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) c[v + k * 256] = r[k] + s * q[k];
    }
  }
  for (; v < v1; v += 256) c[v] = KIND == 4 ? val : (KIND == 2 ? a[v] : a[v] + s * b[v]);
}

}  // namespace

extern "C" int ksmi_hbm_probe(int mode, const void* a, void* b, void* c, size_t nbytes, unsigned* sink, void* stream) {
  if (!a || nbytes < ((size_t)1 << 20) || (nbytes & 15) || !sink) return ksmi_fail(KSMI_E_ARG, "hbm_probe: >= 1 MiB, multiple of 16 bytes, a sink word pair");
  hipStream_t st = (hipStream_t)stream;
  const size_t nvec = nbytes / 16;
  // mode = kind | nt << 3 | grid index << 4: non-temporal loads / stores, 8192 / 4096 / 2048 / 1024 workgroups of the streaming kernels
  const bool nt = (mode >> 3) & 1;
  static const int grids[4] = {8192, 4096, 2048, 1024};
  const int G = grids[(mode >> 4) & 3];
  if (mode & 64) {                                    // contiguous chunk per workgroup; 16 x the workgroups of the strided form
    const int kind = mode & 7, Gc = G * 16;
    if (kind == 2 && b) hipLaunchKernelGGL(probe_chunked<2>, dim3(Gc), dim3(256), 0, st, (const f32x4*)a, (const f32x4*)nullptr, (f32x4*)b, 0.f, nvec);
    else if (kind == 3 && b && c) hipLaunchKernelGGL(probe_chunked<3>, dim3(Gc), dim3(256), 0, st, (const f32x4*)a, (const f32x4*)b, (f32x4*)c, 0.5f, nvec);
    else if (kind == 4) hipLaunchKernelGGL(probe_chunked<4>, dim3(Gc), dim3(256), 0, st, (const f32x4*)nullptr, (const f32x4*)nullptr, (f32x4*)(void*)a, 1.0f, nvec);
    else return ksmi_fail(KSMI_E_ARG, "hbm_probe: the contiguous form has kinds 2, 3, 4");
    return ksmi_check_launch("hbm_probe");
  }
  // mode bits 8..13 (kind 0 only): log2 of the window the LDS-DMA reads wrap in (0 = no wrap: stream all nbytes); bit 14: every
  // workgroup reads the same addresses
  const int wlog = (mode >> 8) & 63, shared = (mode >> 14) & 1;
  mode &= 7;
  switch (mode) {
    case 0: {
      const int grid = 512;                                                  // two workgroups per CU: 16 waves x 8 KB in flight each
      const size_t per = (size_t)1024 * 8 * (kT / 64) * grid;
      if (nbytes % per) return ksmi_fail(KSMI_E_ARG, "hbm_probe: mode 0 needs a multiple of 32 MiB");
      if (wlog && (wlog < 16 || ((size_t)1 << wlog) > nbytes)) return ksmi_fail(KSMI_E_ARG, "hbm_probe: window 64 KiB .. nbytes");
      hipLaunchKernelGGL(probe_read_dma, dim3(grid), dim3(kT), 8192 * (kT / 64), st, (const unsigned char*)a, nbytes, sink,
                         wlog ? (((size_t)1 << wlog) - 1) : (size_t)0, shared);
      break;
    }
    case 1: hipLaunchKernelGGL(probe_read_vec, dim3(G), dim3(256), 0, st, (const u32x4*)a, nvec, sink); break;
    case 2:
      if (!b) return ksmi_fail(KSMI_E_ARG, "hbm_probe: copy needs b");
      if (nt) hipLaunchKernelGGL(probe_copy<true>, dim3(G), dim3(256), 0, st, (const u32x4*)a, (u32x4*)b, nvec);
      else hipLaunchKernelGGL(probe_copy<false>, dim3(G), dim3(256), 0, st, (const u32x4*)a, (u32x4*)b, nvec);
      break;
    case 3:
      if (!b || !c) return ksmi_fail(KSMI_E_ARG, "hbm_probe: triad needs b and c");
      if (nt) hipLaunchKernelGGL(probe_triad<true>, dim3(G), dim3(256), 0, st, (const f32x4*)a, (const f32x4*)b, (f32x4*)c, 0.5f, nvec);
      else hipLaunchKernelGGL(probe_triad<false>, dim3(G), dim3(256), 0, st, (const f32x4*)a, (const f32x4*)b, (f32x4*)c, 0.5f, nvec);
      break;
    case 4:
      if (nt) hipLaunchKernelGGL(probe_fill<true>, dim3(G), dim3(256), 0, st, (f32x4*)(void*)a, 1.0f, nvec);
      else hipLaunchKernelGGL(probe_fill<false>, dim3(G), dim3(256), 0, st, (f32x4*)(void*)a, 1.0f, nvec);
      break;
    default: return ksmi_fail(KSMI_E_ARG, "hbm_probe: kind 0..4");
  }
  return ksmi_check_launch("hbm_probe");
}
