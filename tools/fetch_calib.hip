// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two ways the kernels of this library read HBM (MI355X_MICROARCH.md, HBM
// section: "calibrate on a known byte count in your own access pattern"): a streaming read of N bytes
//   (a) with 16-byte vector loads into registers (the elementwise / reduction kernels),
//   (b) with the LDS-DMA instruction global_load_lds_dwordx4 issued as the convolution kernels issue it (csrc/dma.h glds16_flat:
//       lane i lands at base + 16 i), 1 KB contiguous per wave instruction,
//   (c) as (b) but 64-byte pieces at a 128-byte stride (the first 32-channel chunk of a 64-channel NHWC tensor), N/2 useful bytes.
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o /tmp/fetch_calib
//                         rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fc -o fc -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void glds16_flat(const unsigned char* src, unsigned dst_wave_base) {
  unsigned keep;
  dst_wave_base = __builtin_amdgcn_readfirstlane(dst_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst_wave_base) : "memory");
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(256) void calib_vector_loads(const unsigned char* x, size_t n, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += (size_t)gridDim.x * 256 * 16) {
    const u32x4 v = *(const u32x4*)(x + i);
    acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) *sink = 1;
}

// (d) 1 KB contiguous per wave instruction starting 64 bytes into a line (a halo row that starts at an odd 64-byte pixel): every
//     instruction shares its first and last line with a neighbouring instruction of another wave
// (e) as (b), but the four 16-byte quarters of every 64-byte pixel in permuted order (the bank swizzle applied on the source side)
template <int MODE>
__global__ __launch_bounds__(256) void calib_lds_dma2(const unsigned char* x, size_t n, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned dst = (unsigned)(uintptr_t)&buf[wave][0];
  unsigned acc = 0;
  for (size_t w0 = ((size_t)blockIdx.x * 4 + wave) * 1024; w0 + 2048 <= n; w0 += (size_t)gridDim.x * 4 * 1024) {
    size_t off;
    if (MODE == 0) off = 64 + (size_t)lane * 16;
    else off = (size_t)(lane >> 2) * 64 + (size_t)(((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 16);
    glds16_flat(x + w0 + off, dst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= *(const unsigned*)&buf[wave][lane * 16];
  }
  if (acc == 0x12345678u) *sink = 1;
}

template <int STRIDE128>
__global__ __launch_bounds__(256) void calib_lds_dma(const unsigned char* x, size_t n, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char buf[4][1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned dst = (unsigned)(uintptr_t)&buf[wave][0];
  unsigned acc = 0;
  // one wave instruction: 1 KB contiguous (STRIDE128 = 0) or 16 pieces of 64 B at a 128-byte stride (2 KB span, 1 KB read)
  const size_t span = STRIDE128 ? 2048 : 1024;
  for (size_t w0 = ((size_t)blockIdx.x * 4 + wave) * span; w0 + span <= n; w0 += (size_t)gridDim.x * 4 * span) {
    const size_t off = STRIDE128 ? (size_t)(lane >> 2) * 128 + (lane & 3) * 16 : (size_t)lane * 16;
    glds16_flat(x + w0 + off, dst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= *(const unsigned*)&buf[wave][lane * 16];
  }
  if (acc == 0x12345678u) *sink = 1;
}

int main() {
  const size_t n = (size_t)2 << 30;                  // 2 GiB: far beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
  unsigned char* x; unsigned* sink;
  hipMalloc(&x, n); hipMalloc(&sink, 4);
  hipMemset(x, 1, n); hipMemset(sink, 0, 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(calib_vector_loads, dim3(4096), dim3(256), 0, 0, x, n, sink);
    hipLaunchKernelGGL(calib_lds_dma<0>, dim3(4096), dim3(256), 0, 0, x, n, sink);
    hipLaunchKernelGGL(calib_lds_dma<1>, dim3(4096), dim3(256), 0, 0, x, n, sink);
    hipLaunchKernelGGL(calib_lds_dma2<0>, dim3(4096), dim3(256), 0, 0, x, n, sink);
    hipLaunchKernelGGL(calib_lds_dma2<1>, dim3(4096), dim3(256), 0, 0, x, n, sink);
  }
  hipDeviceSynchronize();
  printf("streamed %zu bytes per launch (kernel c: %zu useful bytes from %zu bytes of lines)\n", n, n / 2, n);
  return 0;
}
