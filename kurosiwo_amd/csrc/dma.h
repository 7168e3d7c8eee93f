// LDS-DMA from inline asm + counted waits + a workgroup barrier that does not drain vector memory (gfx950).
// Shared by the persistent convolution kernels (igemm3.hip, igemm4.hip).
//
// The compiler does not see an asm DMA: it neither drains it at a barrier nor in front of an unrelated global load, so several
// tiles / k-steps stay in flight; completion is waited for with a counted vmcnt.  The counts are exact because every wave issues a
// FIXED number of DMA instructions per stage (positions outside the image read a zero page instead of being exec-masked), and they
// stay safe next to compiler-counted loads and stores: vector-memory reads return in issue order, so "at most n outstanding" can
// only be reached after everything older than the n youngest operations has landed (cdna_hip_programming.md §5, pipelining across
// barriers).
#pragma once
#include <hip/hip_runtime.h>

// lane i of the wave lands at dst_wave_base + 16 * i (LDS byte address, wave-uniform); src is per lane
__device__ __forceinline__ void glds16_flat(const unsigned char* src, unsigned dst_wave_base) {
  unsigned keep;
  dst_wave_base = __builtin_amdgcn_readfirstlane(dst_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst_wave_base) : "memory");
}
// wait until at most n vector-memory operations of this wave are outstanding
__device__ __forceinline__ void vm_wait(int n) {
#define KSMI_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    KSMI_VMW(1) KSMI_VMW(2) KSMI_VMW(3) KSMI_VMW(4) KSMI_VMW(5) KSMI_VMW(6) KSMI_VMW(7) KSMI_VMW(8) KSMI_VMW(9) KSMI_VMW(10)
    KSMI_VMW(11) KSMI_VMW(12) KSMI_VMW(13) KSMI_VMW(14) KSMI_VMW(15) KSMI_VMW(16) KSMI_VMW(17) KSMI_VMW(18) KSMI_VMW(19) KSMI_VMW(20)
    KSMI_VMW(21) KSMI_VMW(22) KSMI_VMW(23) KSMI_VMW(24) KSMI_VMW(25) KSMI_VMW(26) KSMI_VMW(27) KSMI_VMW(28) KSMI_VMW(29) KSMI_VMW(30)
    KSMI_VMW(31) KSMI_VMW(32)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef KSMI_VMW
}
// workgroup barrier that does NOT drain vector memory: LDS operations of this wave complete, then s_barrier
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
