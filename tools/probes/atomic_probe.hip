// Probe: cost of finishing per-workgroup statistics with 64-bit integer atomics (exact, order-independent) instead of partial rows
// + a finalize launch.  Each of G workgroups adds V values (two u64 atomics per value: lo with return -> carry -> hi) into row
// (wg % R) of a [R][V][2] u64 accumulator.  Build: hipcc -O3 --offload-arch=gfx950 tools/probes/atomic_probe.hip -o /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void body(float* sink, int spin) {           // stand-in for the producer's work: a few microseconds of ALU
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) sink[0] = v;
}
__global__ void atom(unsigned long long* acc, int V, int R, float* sink, int spin, int mode) {
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) sink[0] = v;
  const int row = blockIdx.x % R;
  for (int j = threadIdx.x; j < V; j += blockDim.x) {
    const double x = 1.0 + 1e-3 * j + 1e-7 * blockIdx.x;
    const double fl = floor(x);
    const long long hi = (long long)fl;
    const unsigned long long lo = (unsigned long long)((x - fl) * 18446744073709551616.0);
    unsigned long long* p = acc + ((size_t)row * V + j) * 2;
    if (mode == 0) {
      const unsigned long long old = atomicAdd(p, lo);
      const unsigned long long carry = (old + lo) < old ? 1ull : 0ull;
      atomicAdd(p + 1, (unsigned long long)hi + carry);
    } else if (mode == 1) {
      atomicAdd((double*)p, x);                        // fp64 atomic, no return
    } else {
      ((double*)p)[0] = x;                             // plain store (row per wg would be the real layout): baseline
    }
  }
}
int main() {
  const int Gs[] = {256, 512, 2048, 6272}, Vs[] = {64, 256, 1024}, Rs[] = {1, 4, 16};
  unsigned long long* acc; float* sink;
  hipMalloc(&acc, 16 * 1024 * 2 * 8 * 2); hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int G : Gs) for (int V : Vs) for (int R : Rs) {
      if (mode == 2 && R != 1) continue;
      hipMemset(acc, 0, 16 * 1024 * 2 * 8 * 2);
      for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(atom, dim3(G), dim3(256), 0, 0, acc, V, R, sink, 2000, mode);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(atom, dim3(G), dim3(256), 0, 0, acc, V, R, sink, 2000, mode);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s) G %5d V %5d R %2d : %.2f us per launch\n", mode, mode == 0 ? "u64 pair" : mode == 1 ? "f64 atomic" : "store", G, V, R, ms * 1000 / 20);
    }
  // reference: the same body without any tail
  for (int G : Gs) {
    hipEventRecord(e0);
    for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(body, dim3(G), dim3(256), 0, 0, sink, 2000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("body only G %5d : %.2f us per launch\n", G, ms * 1000 / 20);
  }
  // exactness check of the pair accumulation (mode 0, G=512, V=64, R=1)
  hipMemset(acc, 0, 16 * 1024 * 2 * 8 * 2);
  hipLaunchKernelGGL(atom, dim3(512), dim3(256), 0, 0, acc, 64, 1, sink, 10, 0);
  std::vector<unsigned long long> h(64 * 2);
  hipMemcpy(h.data(), acc, 64 * 2 * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int j = 0; j < 64; ++j) {
    long double ref = 0;
    for (int b = 0; b < 512; ++b) ref += (long double)(1.0 + 1e-3 * j + 1e-7 * b);
    const long double got = (long double)(long long)h[2 * j + 1] + (long double)h[2 * j] / 18446744073709551616.0L;
    const double rel = (double)fabsl((got - ref) / ref);
    if (rel > worst) worst = rel;
  }
  printf("pair accumulation worst rel err vs long double: %.3g\n", worst);
  return 0;
}
