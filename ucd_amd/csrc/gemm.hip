// bf16 row-major GEMMs of the wide 1x1 convolutions (forward, input gradient, weight gradient) through hipBLASLt,
// with the algorithm chosen ONCE per shape by timing the library's candidates (the same idea as MIOpen's solver search
// for the spatial convolutions) and cached: the framework path pays ~18 us of host time per call for the heuristic
// query and takes its first suggestion - which for the weight gradient dW = dY^T X (K = B*H*W = 26136) is a 16-workgroup
// kernel on a 256-CU part.  These are plain library GEMMs (no fusion), so they stay on hipBLASLt.
//
// hipBLASLt is bound at run time from the copy already loaded in the process (PyTorch-ROCm ships its own).
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace ucd {
namespace {

struct LtApi {
  void* lib = nullptr;
  hipblasLtHandle_t handle = nullptr;
  decltype(&hipblasLtCreate) Create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
  decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
  decltype(&hipblasLtMatmul) Matmul = nullptr;
} g_lt;

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  float tuned_us = -1.f;
  int candidates = 0;
};

typedef std::tuple<int, int, int, int, int, int, int> Key;   // mode, M, N, K, lda, ldb, ldc
std::map<Key, Plan> g_plans;
std::mutex g_mu;
float g_last_us = -1.f;
int g_last_candidates = 0;

constexpr size_t kWorkspace = (size_t)64 << 20;

#define UCD_LT(fn, call)                                                 \
  do {                                                                   \
    hipblasStatus_t st_ = (call);                                        \
    if (st_ != HIPBLAS_STATUS_SUCCESS) {                                 \
      set_error("%s: hipBLASLt status %d", fn, (int)st_);                \
      return UCD_EBLAS_BASE + (int)st_;                                  \
    }                                                                    \
  } while (0)

int launch(const Plan& p, const void* ltA, const void* ltB, void* C, void* ws, hipStream_t s, float beta = 0.f) {
  const float alpha = 1.f;
  UCD_LT("hipblasLtMatmul", g_lt.Matmul(g_lt.handle, p.desc, &alpha, ltA, p.la, ltB, p.lb, &beta, C, p.lc, C, p.lc, &p.algo, ws,
                                        p.workspace, s));
  return 0;
}

int make_plan(Plan& p, int mode, int M, int N, int K, int lda, int ldb, int ldc, const void* ltA, const void* ltB, void* C, void* ws,
              size_t ws_bytes, int tune, hipStream_t s) {
  static const char* fn = "ucd_gemm_bf16";
  // column-major restatement: C^T[N, M] = op(ltA)[N, K] . op(ltB)[K, M]
  const hipblasOperation_t opA = mode == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  const hipblasOperation_t opB = mode == 2 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  UCD_LT(fn, g_lt.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  UCD_LT(fn, g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opA, sizeof(int32_t)));
  UCD_LT(fn, g_lt.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opB, sizeof(int32_t)));
  // stored (column-major) shapes of the two operands
  if (mode == 0) UCD_LT(fn, g_lt.LayoutCreate(&p.la, HIP_R_16BF, K, N, ldb));       // B[N,K] row-major = [K,N] col-major, transposed
  else UCD_LT(fn, g_lt.LayoutCreate(&p.la, HIP_R_16BF, N, K, ldb));                 // B[K,N] row-major = [N,K] col-major
  if (mode == 2) UCD_LT(fn, g_lt.LayoutCreate(&p.lb, HIP_R_16BF, M, K, lda));       // A[K,M] row-major = [M,K] col-major, transposed
  else UCD_LT(fn, g_lt.LayoutCreate(&p.lb, HIP_R_16BF, K, M, lda));                 // A[M,K] row-major = [K,M] col-major
  UCD_LT(fn, g_lt.LayoutCreate(&p.lc, HIP_R_16BF, N, M, ldc));
  hipblasLtMatmulPreference_t pref = nullptr;
  UCD_LT(fn, g_lt.PrefCreate(&pref));
  uint64_t maxws = ws ? ws_bytes : 0;
  UCD_LT(fn, g_lt.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &maxws, sizeof maxws));
  constexpr int kReq = 24;
  std::vector<hipblasLtMatmulHeuristicResult_t> res(kReq);
  int got = 0;
  UCD_LT(fn, g_lt.Heuristic(g_lt.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, tune ? kReq : 1, res.data(), &got));
  UCD_REQUIRE(got > 0, UCD_EUNSUPPORTED, "%s: hipBLASLt has no algorithm for mode %d M=%d N=%d K=%d", fn, mode, M, N, K);
  p.candidates = got;
  int best = 0;
  if (tune && got > 1) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return check_launch(fn);
    float best_ms = 1e30f;
    for (int i = 0; i < got; ++i) {
      if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > maxws) continue;
      Plan trial = p;
      trial.algo = res[i].algo;
      trial.workspace = res[i].workspaceSize;
      bool ok = true;
      for (int w = 0; w < 2 && ok; ++w) ok = launch(trial, ltA, ltB, C, ws, s) == 0;
      if (!ok) continue;
      float ms = 1e30f;
      for (int round = 0; round < 2; ++round) {      // best of two timed rounds: one noisy round must not pick the plan
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < 5; ++r) (void)launch(trial, ltA, ltB, C, ws, s);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); ms = 1e30f; break; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        if (t < ms) ms = t;
      }
      if (ms > 1e29f) continue;
      if (ms < best_ms) { best_ms = ms; best = i; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    p.tuned_us = best_ms < 1e29f ? best_ms * 1000.f / 5.f : -1.f;
  }
  p.algo = res[best].algo;
  p.workspace = res[best].workspaceSize;
  return 0;
}

}  // namespace
}  // namespace ucd

using namespace ucd;

extern "C" {

int ucd_gemm_load(const char* path) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_lt.handle) return 0;
  const char* candidates[] = {path, "libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so"};
  void* h = nullptr;
  for (const char* c : candidates) {
    if (!c || !*c) continue;
    h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  UCD_REQUIRE(h, UCD_EINVAL, "ucd_gemm_load: cannot open hipBLASLt (%s)", dlerror());
#define BIND(field, sym)                                                          \
  g_lt.field = reinterpret_cast<decltype(g_lt.field)>(dlsym(h, sym));             \
  UCD_REQUIRE(g_lt.field, UCD_EINVAL, "ucd_gemm_load: symbol %s not found", sym)
  BIND(Create, "hipblasLtCreate");
  BIND(DescCreate, "hipblasLtMatmulDescCreate");
  BIND(DescSet, "hipblasLtMatmulDescSetAttribute");
  BIND(LayoutCreate, "hipblasLtMatrixLayoutCreate");
  BIND(PrefCreate, "hipblasLtMatmulPreferenceCreate");
  BIND(PrefSet, "hipblasLtMatmulPreferenceSetAttribute");
  BIND(Heuristic, "hipblasLtMatmulAlgoGetHeuristic");
  BIND(Matmul, "hipblasLtMatmul");
#undef BIND
  g_lt.lib = h;
  UCD_LT("hipblasLtCreate", g_lt.Create(&g_lt.handle));
  return 0;
}

size_t ucd_gemm_workspace_bytes(void) { return kWorkspace; }

static int gemm_impl(const char* fn, int mode, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                     int ldc, void* workspace, size_t workspace_bytes, int tune, float beta, ucd_stream_t stream) {
  UCD_REQUIRE(g_lt.handle, UCD_EINVAL, "%s: call ucd_gemm_load first", fn);
  UCD_REQUIRE(mode >= 0 && mode <= 2 && M > 0 && N > 0 && K > 0 && A && B && C, UCD_EINVAL, "%s: bad arguments", fn);
  UCD_REQUIRE(lda >= (mode == 2 ? M : K) && ldb >= (mode == 0 ? K : N) && ldc >= N, UCD_EINVAL, "%s: leading dimension too small", fn);
  hipStream_t s = (hipStream_t)stream;
  const void* ltA = B;   // see make_plan: hipBLASLt's A operand is our B
  const void* ltB = A;
  Plan* plan;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    Key key(mode, M, N, K, lda, ldb, ldc);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
      Plan p;
      int rc = make_plan(p, mode, M, N, K, lda, ldb, ldc, ltA, ltB, C, workspace, workspace_bytes, tune, s);
      if (rc) return rc;
      it = g_plans.emplace(key, p).first;
    }
    plan = &it->second;
    g_last_us = plan->tuned_us;
    g_last_candidates = plan->candidates;
  }
  UCD_REQUIRE(plan->workspace <= workspace_bytes || plan->workspace == 0, UCD_EWORKSPACE, "%s: workspace too small", fn);
  return launch(*plan, ltA, ltB, C, workspace, s, beta);
}

int ucd_gemm_bf16(int mode, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                  void* workspace, size_t workspace_bytes, int tune, ucd_stream_t stream) {
  return gemm_impl("ucd_gemm_bf16", mode, M, N, K, A, lda, B, ldb, C, ldc, workspace, workspace_bytes, tune, 0.f, stream);
}

/* C += op(A) op(B): same plan (algorithm) as ucd_gemm_bf16 for the shape; never tunes (tuning relaunches, which would
 * accumulate several times) - warm the shape with ucd_gemm_bf16(tune = 1) into a scratch matrix first */
int ucd_gemm_bf16_acc(int mode, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                      void* workspace, size_t workspace_bytes, ucd_stream_t stream) {
  return gemm_impl("ucd_gemm_bf16_acc", mode, M, N, K, A, lda, B, ldb, C, ldc, workspace, workspace_bytes, 0, 1.f, stream);
}

int ucd_gemm_has_plan(int mode, int M, int N, int K, int lda, int ldb, int ldc) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_plans.count(Key(mode, M, N, K, lda, ldb, ldc)) ? 1 : 0;
}

/* introspection for tools/tests: tuned duration (us, -1 if not tuned) and number of candidates of the last plan used */
float ucd_gemm_last_tuned_us(void) { return g_last_us; }
int ucd_gemm_last_candidates(void) { return g_last_candidates; }

}  // extern "C"
