// thread-local error reporting shared by all translation units of libksmi
#pragma once
#include <hip/hip_runtime.h>
int ksmi_fail(int code, const char* msg);          // records msg, returns code
int ksmi_check_launch(const char* what);           // hipGetLastError() -> 0 or positive hipError_t

// igemm2.hip: software-pipelined implicit-GEMM (LDS-DMA double buffering)
struct ksmi_conv_desc;
bool ksmi_igemm2_eligible(const ksmi_conv_desc* d, int dtype);
int ksmi_igemm2_launch(const ksmi_conv_desc* d, int dtype, hipStream_t st);
