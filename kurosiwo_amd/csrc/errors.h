// thread-local error reporting shared by all translation units of libksmi
#pragma once
#include <hip/hip_runtime.h>
int ksmi_fail(int code, const char* msg);          // records msg, returns code
int ksmi_check_launch(const char* what);           // hipGetLastError() -> 0 or positive hipError_t
