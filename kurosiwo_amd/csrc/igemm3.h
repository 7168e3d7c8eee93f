// igemm3.hip: persistent implicit-GEMM convolution with register-resident weights (short K, bf16).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ksmi.h"

struct ksmi_igemm3_geom_t {
  int WN, NTI;          // channel groups per workgroup (32 columns each); column tiles walked inside a workgroup
  int hpb, nslot, stage, ns;
  int tiles, gx, gy;    // pixel tiles; grid (gx persistent workgroups along the pixel axis = rows of `stats`)
  size_t lds;
};
// false: the descriptor does not qualify (the caller uses igemm2)
bool ksmi_igemm3_geom(const ksmi_conv_desc* d, int dtype, ksmi_igemm3_geom_t* g);
int ksmi_igemm3_launch(const ksmi_conv_desc* d, const ksmi_igemm3_geom_t* g, hipStream_t st);
