// NOT part of libksmi.so since round 3 (the product library holds hand-written kernels only): kept as the record of the round-1/2
// hipBLASLt comparison route.  The library-vs-hand-written comparison is profiles/gemm_probe.py (torch.matmul = hipBLASLt).
// OPTIONAL comparison path, off by default: plain nn.Linear GEMMs (vision_transformer.py:22-31,47-50; changeformer.py:110-113,157-161)
// above a size threshold can be sent to hipBLASLt with KSMI_USE_HIPBLASLT=1, to time the library next to the hand-written
// LDS-DMA kernels of gemm2.hip, which are the product path (FloodViT step, MI355X: 949 tiles/s hand-written, 976 with the
// library's 256-wide stream-K tiles; profiles/gemm_probe.py, tools/gemm_durations.py).
//
// The library is bound at run time (dlopen + dlsym; the copy a host process already holds is reused), so libksmi.so has no link
// dependency on it: without KSMI_USE_HIPBLASLT=1 (or without the library) every call below returns "not taken" and the caller runs its own
// kernel.  Row-major operands are passed as their column-major transposes:
//   forward      Y^T [N x rows] = W [N x K] X^T      -> op(A) = T on the stored K x N image of W, op(B) = N on X (K x rows)
//   input grad   dX^T [K x rows] = W^T dY^T          -> op(A) = N on W (K x N),                 op(B) = N on dY (N x rows)
//   weight grad  dW^T [K x N]    = X^T dY            -> op(A) = N on X (K x rows),              op(B) = T on dY (N x rows)
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include <hipblaslt/hipblaslt.h>
#include "common.h"
#include "../../include/ksmi.h"
#include "errors.h"

namespace {

struct LtApi {
  decltype(&hipblasLtCreate) Create;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate;
  decltype(&hipblasLtMatmulDescCreate) DescCreate;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic;
  decltype(&hipblasLtMatmul) Matmul;
  hipblasLtHandle_t handle;
  hipblasLtMatmulPreference_t pref;
  void* workspace;
  size_t ws_bytes;
};

LtApi* lt_api() {
  static LtApi api;
  static int state = 0;                      // 0 untried, 1 ready, -1 unavailable
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (state) return state > 0 ? &api : nullptr;
  state = -1;
  if (!getenv("KSMI_USE_HIPBLASLT") || getenv("KSMI_NO_HIPBLASLT")) return nullptr;
  void* h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/libhipblaslt.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return nullptr;
#define KSMI_LT_SYM(field, name)                                        \
  api.field = (decltype(api.field))dlsym(h, name);                      \
  if (!api.field) return nullptr
  KSMI_LT_SYM(Create, "hipblasLtCreate");
  KSMI_LT_SYM(LayoutCreate, "hipblasLtMatrixLayoutCreate");
  KSMI_LT_SYM(DescCreate, "hipblasLtMatmulDescCreate");
  KSMI_LT_SYM(DescSet, "hipblasLtMatmulDescSetAttribute");
  KSMI_LT_SYM(PrefCreate, "hipblasLtMatmulPreferenceCreate");
  KSMI_LT_SYM(PrefSet, "hipblasLtMatmulPreferenceSetAttribute");
  KSMI_LT_SYM(Heuristic, "hipblasLtMatmulAlgoGetHeuristic");
  KSMI_LT_SYM(Matmul, "hipblasLtMatmul");
#undef KSMI_LT_SYM
  if (api.Create(&api.handle) != HIPBLAS_STATUS_SUCCESS) return nullptr;
  api.ws_bytes = (size_t)64 << 20;
  if (hipMalloc(&api.workspace, api.ws_bytes) != hipSuccess) return nullptr;
  if (api.PrefCreate(&api.pref) != HIPBLAS_STATUS_SUCCESS) return nullptr;
  uint64_t ws = api.ws_bytes;
  if (api.PrefSet(api.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)) != HIPBLAS_STATUS_SUCCESS) return nullptr;
  state = 1;
  return &api;
}

// opA, opB, m, n, k, lda, ldb, ldc, ldd, C/D fp32?, bias?
typedef std::tuple<int, int, int, int, int, int, int, int, int, int, int> LtKey;
struct LtPlan {
  bool ok;
  hipblasLtMatmulDesc_t desc;
  hipblasLtMatrixLayout_t la, lb, lc, ld;
  hipblasLtMatmulAlgo_t algo;
};

// D (m x n, column major, ldd) = op(A) op(B) + beta C (+ bias[m]); A, B bf16; C, D bf16 or fp32; fp32 accumulation.
// Returns 0 when the product was launched, 1 when hipBLASLt is absent or has no kernel for the problem (caller falls back).
int lt_gemm(bool ta, bool tb, int m, int n, int k, const void* A, int lda, const void* B, int ldb, const void* C, int ldc, void* D, int ldd,
            bool f32out, float beta, const float* bias, hipStream_t st) {
  LtApi* api = lt_api();
  if (!api) return 1;
  static std::map<LtKey, LtPlan> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  const LtKey key(ta, tb, m, n, k, lda, ldb, ldc, ldd, f32out, bias != nullptr);
  auto it = cache.find(key);
  if (it == cache.end()) {
    LtPlan p;
    p.ok = false;
    const hipDataType ab = HIP_R_16BF, cd = f32out ? HIP_R_32F : HIP_R_16BF;
    bool good = api->DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    const int32_t opa = ta ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = tb ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    good = good && api->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)) == HIPBLAS_STATUS_SUCCESS;
    good = good && api->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)) == HIPBLAS_STATUS_SUCCESS;
    if (good && bias) {
      const uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;
      const int32_t bt = HIP_R_32F;
      good = api->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS &&
             api->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) == HIPBLAS_STATUS_SUCCESS &&
             api->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) == HIPBLAS_STATUS_SUCCESS;
    }
    // stored shapes (column major): A is (ta ? k x m : m x k), B is (tb ? n x k : k x n)
    good = good && api->LayoutCreate(&p.la, ab, ta ? k : m, ta ? m : k, lda) == HIPBLAS_STATUS_SUCCESS;
    good = good && api->LayoutCreate(&p.lb, ab, tb ? n : k, tb ? k : n, ldb) == HIPBLAS_STATUS_SUCCESS;
    good = good && api->LayoutCreate(&p.lc, cd, m, n, ldc) == HIPBLAS_STATUS_SUCCESS;
    good = good && api->LayoutCreate(&p.ld, cd, m, n, ldd) == HIPBLAS_STATUS_SUCCESS;
    if (good) {
      hipblasLtMatmulHeuristicResult_t res[1];
      int found = 0;
      if (api->Heuristic(api->handle, p.desc, p.la, p.lb, p.lc, p.ld, api->pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS && found > 0 &&
          res[0].workspaceSize <= api->ws_bytes) {
        p.algo = res[0].algo;
        p.ok = true;
      }
    }
    if (!p.ok && getenv("KSMI_LT_DEBUG"))
      fprintf(stderr, "ksmi: hipBLASLt has no kernel for ta=%d tb=%d m=%d n=%d k=%d lda=%d ldb=%d ldc=%d ldd=%d f32out=%d bias=%d (layouts ok=%d)\n", (int)ta,
              (int)tb, m, n, k, lda, ldb, ldc, ldd, (int)f32out, bias != nullptr, (int)good);
    it = cache.emplace(key, p).first;
  }
  LtPlan& p = it->second;
  if (!p.ok) return 1;
  if (bias && api->DescSet(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS) return 1;
  const float alpha = 1.f;
  const hipblasStatus_t rc = api->Matmul(api->handle, p.desc, &alpha, A, p.la, B, p.lb, &beta, C, p.lc, D, p.ld, &p.algo, api->workspace,
                                         api->ws_bytes, st);
  if (rc != HIPBLAS_STATUS_SUCCESS) {
    if (getenv("KSMI_LT_DEBUG")) fprintf(stderr, "ksmi: hipblasLtMatmul failed with status %d (m=%d n=%d k=%d f32out=%d)\n", (int)rc, m, n, k, (int)f32out);
    p.ok = false;
    return 1;
  }
  return 0;
}

// problems below this many GFLOP stay on the hand-written tiles (they win there: profiles/gemm_probe.py)
double lt_min_gflop() {
  static const double v = getenv("KSMI_LT_MIN_GFLOP") ? atof(getenv("KSMI_LT_MIN_GFLOP")) : 5.0;
  return v;
}
// ... and so do thin problems whatever their size (ChangeFormer's 64..320-wide token GEMMs over 2 x 10^5 rows: 9.6 us hand-written vs 19.8)
bool lt_wanted(int rows, int K, int N) { return K >= 512 && N >= 512 && 2.0 * rows * K * N >= lt_min_gflop() * 1e9; }

}  // namespace

// ---- hooks used by gemm.hip / igemm.hip: 0 = launched, 1 = not taken -----------------------------------------------
int ksmi_lt_linear_forward(const void* x, int x_rs, const void* w, int w_rs, const float* bias, const void* resid, int r_rs, void* y, int y_rs,
                           int rows, int K, int N, hipStream_t st) {
  if (!lt_wanted(rows, K, N)) return 1;
  return lt_gemm(true, false, N, rows, K, w, w_rs, x, x_rs, resid ? resid : y, resid ? r_rs : y_rs, y, y_rs, false, resid ? 1.f : 0.f, bias, st);
}

int ksmi_lt_linear_dgrad(const void* dy, int dy_rs, const void* w, int w_rs, void* dx, int dx_rs, int rows, int K, int N, int accumulate,
                         hipStream_t st) {
  if (!lt_wanted(rows, K, N)) return 1;
  return lt_gemm(false, false, K, rows, N, w, w_rs, dy, dy_rs, dx, dx_rs, dx, dx_rs, false, accumulate ? 1.f : 0.f, nullptr, st);
}

int ksmi_lt_linear_wgrad(const void* x, int x_rs, const void* dy, int dy_rs, float* grad, int g_rs, int rows, int K, int N, int accumulate,
                         hipStream_t st) {
  if (!lt_wanted(rows, K, N)) return 1;
  return lt_gemm(false, true, K, N, rows, x, x_rs, dy, dy_rs, grad, g_rs, grad, g_rs, true, accumulate ? 1.f : 0.f, nullptr, st);
}
