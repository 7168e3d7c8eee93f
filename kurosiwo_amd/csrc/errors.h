// thread-local error reporting shared by all translation units of libksmi
#pragma once
#include <hip/hip_runtime.h>
int ksmi_fail(int code, const char* msg);          // records msg, returns code
int ksmi_check_launch(const char* what);           // hipGetLastError() -> 0 or positive hipError_t
// run-time knobs (api.hip): the environment variable `name` as read at the knob's FIRST look-up, or what ksmi_set_knob stored since
int ksmi_knob_int(const char* name, int dflt);
bool ksmi_knob_str(const char* name, char* out, int cap);      // false: unset
bool ksmi_knob_is_set(const char* name);
// (every launcher switch goes through these: the start-up switches as `static const` reads -- evaluated once, at first use --, the
// run-time knobs of the tests per call; no translation unit calls getenv itself)

// Address of a __device__ symbol (the zero pages the LDS-DMA loaders read padding from) on the CURRENT device, cached per device:
// a process that drives several GPUs must not hand device 1 the address device 0 resolved.  Concurrent first calls
// resolve the same value twice at worst.  Defines `static const unsigned char* fn()`; nullptr = lookup failed.
#define KSMI_MAX_DEVICES 64
#define KSMI_DEVICE_SYMBOL_GETTER(fn, sym)                                                                          \
  static const unsigned char* fn() {                                                                                \
    static void* cache[KSMI_MAX_DEVICES];                                                                           \
    int dev = 0;                                                                                                    \
    if (hipGetDevice(&dev) != hipSuccess || (unsigned)dev >= KSMI_MAX_DEVICES) return nullptr;                      \
    void* p = __atomic_load_n(&cache[dev], __ATOMIC_ACQUIRE);                                                       \
    if (!p) {                                                                                                       \
      if (hipGetSymbolAddress(&p, HIP_SYMBOL(sym)) != hipSuccess) return nullptr;                                   \
      __atomic_store_n(&cache[dev], p, __ATOMIC_RELEASE);                                                           \
    }                                                                                                               \
    return (const unsigned char*)p;                                                                                 \
  }

// igemm2.hip: software-pipelined implicit-GEMM (LDS-DMA double buffering)
struct ksmi_conv_desc;
bool ksmi_igemm2_eligible(const ksmi_conv_desc* d, int dtype);
int ksmi_igemm2_launch(const ksmi_conv_desc* d, int dtype, hipStream_t st);

// attn_mfma.hip: MFMA attention (bf16)
int ksmi_attn_mfma_qsplit(int B, int Nq, int Nk, int H);
size_t ksmi_attn_mfma_workspace(int B, int Nq, int Nk, int H, int D);
int ksmi_attn_mfma_vit(int backward, const void* qkv, void* out, float* lse, const void* dout, void* dqkv, void* workspace, int B, int N,
                       int H, float scale, void* stream);
int ksmi_attn_mfma_sr(int backward, const void* q, const void* kv, void* out, float* lse, const void* dout, void* dq, void* dkv,
                      void* workspace, int B, int Nq, int Nk, int H, int C, float scale, uint32_t drop_thr, float drop_inv,
                      uint32_t drop_site, const uint32_t* rng, void* stream);

// gemm2.hip: LDS-DMA token GEMMs; 0 = launched, 1 = shape not covered (fall back to gemm.hip), < 0 = error
int ksmi_gemm2_nt(const void* x, int x_rs, const void* w, int w_rs, const float* bias, const void* resid, int r_rs, void* y, int y_rs,
                  int rows, int K, int N, hipStream_t st);
int ksmi_gemm2_nn(const void* dy, int dy_rs, const void* w, int w_rs, void* dx, int dx_rs, int rows, int K, int N, int accumulate,
                  hipStream_t st);
int ksmi_gemm2_tn(const void* x, int x_rs, const void* dy, int dy_rs, float* slab, int npad, float* grad, int64_t g_rs, int rows, int K, int N,
                  int Kslab, int nsplit, int rows_per_split, int btile, int accumulate, float* bias_grad, int bias_accumulate, hipStream_t st);
bool ksmi_gemm2_tn_enabled(int K, int N, int rows_per_split);
bool ksmi_gemm2_tn_spl(int tiles_times_splits, int steps_per_split);     // gemm2_tn_kernel runs two wave groups per workgroup (SPL = 2)
