// wgrad3.hip: weight gradient of 3x3 / stride 1 / pad 1 convolutions (bf16), channel-owner tiling.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ksmi.h"

struct ksmi_wgrad3_geom_t {
  int TH, TW, HWc, tilesX, tilesY;   // pixel patch (TH*TW <= 128), staged halo width, patches per image
  int WC, NTL;                       // waves along the channel axis (2 / 4 -> 32 / 64 channels per tile), columns per tile (32 / 64)
  int KT, NTt;                       // tiles along K and N
  int patches, pps, nsplit;          // patches, patches per split, splits (= partial slabs)
  int xpl, stage;                    // LDS bytes of one X plane / one stage
  int nst;                           // stages of the LDS ring (2: two workgroups per CU; up to 4 when the workgroup owns the CU)
  int r0, c0;                        // >= 0: 2 x 2 window of taps (origin inside the 3 x 3 neighbourhood) of a phase gradient; -1: all 9 taps
  size_t lds;
};
// false: the descriptor does not qualify (the caller uses igemm_wgrad_kernel)
bool ksmi_wgrad3_geom(const ksmi_wgrad_desc* d, int dtype, ksmi_wgrad3_geom_t* g);
int ksmi_wgrad3_launch(const ksmi_wgrad_desc* d, const ksmi_wgrad3_geom_t* g, hipStream_t st);
int ksmi_wgrad3_reduce(const ksmi_wgrad_desc* d, const ksmi_wgrad3_geom_t* g, hipStream_t st);   // slabs -> fp32 gradient
