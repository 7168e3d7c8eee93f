// igemm4.hip: persistent implicit-GEMM 3x3 convolution for LONG K (>= 2 k-chunks) and >= 64 output channels, bf16:
// halo ring + weight ring filled by asm LDS-DMA with counted vmcnt, one raw barrier per kernel row of taps.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ksmi.h"

struct ksmi_igemm4_geom_t {
  int WM, NF;           // pixel groups of 64 (4 | 8; column groups = nwv / WM); 16-column MFMA fragments per wave (2 | 4)
  int nwv;              // waves per workgroup: 8 (one workgroup per CU) | 4 (two per CU, round 5)
  int deep;             // 1: 5-slot weight ring, weights four steps ahead (run_tiles_deep); 2: one barrier per chunk (run_tiles_chunk); round 5
  int th, tw;           // output patch of a workgroup (the descriptor's for WM = 4, chosen by the kernel for WM = 8)
  int hslot, nh, nhs;   // halo ring: bytes per slot, DMA pieces per wave and slot, slots
  int tiles, gx, gy;    // pixel tiles; persistent workgroups along the pixel axis (= rows of `stats`); column tiles
  size_t lds;
};
// false: the descriptor does not qualify (the caller uses igemm2)
bool ksmi_igemm4_geom(const ksmi_conv_desc* d, int dtype, ksmi_igemm4_geom_t* g);
int ksmi_igemm4_launch(const ksmi_conv_desc* d, const ksmi_igemm4_geom_t* g, hipStream_t st);
// true: ksmi_igemm4_launch has an instantiated kernel for this geometry (host only, nothing is launched; ksmi_conv_dispatch_info)
bool ksmi_igemm4_launchable(const ksmi_conv_desc* d, const ksmi_igemm4_geom_t* g);
