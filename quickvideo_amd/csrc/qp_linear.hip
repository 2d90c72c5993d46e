// out = act(alpha * x W^T + b) for a bf16 Linear through hipBLASLt with the activation in the GEMM EPILOGUE (plain library GEMM, no
// own kernel).  The vision tower's fc1 + quick-GELU (transformers VisionMlp [3P]: QuickGELUActivation, y * sigmoid(1.702 y)) becomes
// ONE GEMM: with alpha = 1.702 and the bias pre-scaled by 1.702 the Swish epilogue (z * sigmoid(z)) yields 1.702 * quick_gelu(y), and
// the factor 1/1.702 rides on the alpha of the fc2 GEMM — the [n, 5120] intermediate is never re-read and re-written by a separate
// activation kernel.  fp32 accumulation + bias + activation, ONE rounding to bf16 (the unfused path rounds after the bias and
// inside the activation: results agree to a bf16 ulp or two).  (The process runs torch's bundled hipBLASLt, ROCm 7.0: its
// Swish epilogue has no slope argument, hence the alpha route.)
// Row-major x [m][k], W [n][k] (a torch Linear weight), out [m][n]  ==  column-major  D[n x m] = op_T(W[k x n]) * x[k x m].
#include "qp_common.h"
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  int pick = 0;                                                // index into cands of the algorithm in use
  std::vector<hipblasLtMatmulHeuristicResult_t> cands;         // every heuristic candidate (qp_linear_tune picks among them)
};

struct LtState {
  // One hipBLASLt handle PER STREAM: the library keeps device-side state per handle (the flag / partial-tile buffer of its stream-K
  // kernels), so GEMMs of two streams that run concurrently (ViT stream and prefill stream, pipeline.py) must not go through the
  // same handle — with one shared handle the device stalls for good in the video -> first-token leg (tools/repro_pipeline.py).
  // The state lives in the qp_ctx, i.e. per DEVICE: stream 0 exists on every device, and a handle (or a plan tuned on one device's
  // handle) must not be reused on another.  qp_destroy frees handles, descriptors and layouts.
  std::map<hipStream_t, hipblasLtHandle_t> handles;
  std::map<std::tuple<int64_t, int64_t, int64_t, int, int>, Plan> plans;   // (m, n, k, act, bias kind: 0 none, 1 bf16, 2 fp32)
  std::mutex mu;
};

std::mutex g_create_mu;

// The tuner's decisions are per (DEVICE, problem) and belong to the PROCESS, not to one context: the host side remembers "this shape is
// tuned" per device (engine.py: QuickPrefillEngine._SHARED), and every context created later on the same device must run the algorithm
// the stopwatch picked, not fall back to heuristic candidate 0.  Rounds 3-4 kept the choice in the context's plan only: a second engine
// of the process (own qp_ctx) then skipped the tuning — its shape was on record as tuned — and ran candidate 0, a different fp32
// accumulation order whenever the first context's tuner had picked another candidate.  That was the "one in ~15 suite runs" rounding
// mismatch between the one-call and the per-operator path in tests/test_gpu_engine.py: two engines, two contexts, and a
// first pick != 0 on a noise-dominated tiny shape.  One table for all contexts: a problem is timed ONCE per device and process.
// What is recorded is the ALGORITHM (hipBLASLt's 16-byte algo blob + its workspace need), not its position in a candidate list: the
// list hipblasLtMatmulAlgoGetHeuristic returns depends on the workspace limit and the handle it was asked with, so the same index can
// name different algorithms in two contexts.  A context adopts the record only when its own list holds that very algorithm.
struct Choice {
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
};
std::mutex g_choice_mu;
std::map<std::tuple<int, int64_t, int64_t, int64_t, int, int>, Choice> g_choice;   // (device, m, n, k, act, bias kind) -> tuned algorithm

bool recorded_choice(int device, int64_t m, int64_t n, int64_t k, int act, int bias_kind, Choice* out) {
  std::lock_guard<std::mutex> g(g_choice_mu);
  auto it = g_choice.find(std::make_tuple(device, m, n, k, act, bias_kind));
  if (it == g_choice.end()) return false;
  if (out) *out = it->second;
  return true;
}

bool same_algo(const hipblasLtMatmulAlgo_t& a, const hipblasLtMatmulAlgo_t& b) { return memcmp(a.data, b.data, sizeof(a.data)) == 0; }

// position of the recorded algorithm in this context's candidate list, -1 if the list does not hold it (or it needs more workspace
// than this caller offers)
int find_choice(const std::vector<hipblasLtMatmulHeuristicResult_t>& cands, const Choice& c, size_t workspace_bytes) {
  for (int i = 0; i < (int)cands.size(); ++i)
    if (same_algo(cands[i].algo, c.algo)) return cands[i].workspaceSize <= workspace_bytes ? i : -1;
  return -1;
}

#ifdef QP_EXPERIMENTS
const char* env_lt_algo_index() { static const char* e = getenv("QP_LT_ALGO_INDEX"); return e; }
#endif
bool env_lt_debug() { static const bool d = getenv("QP_LT_DEBUG") != nullptr; return d; }

LtState& lt(qp_ctx* ctx) {
  std::lock_guard<std::mutex> g(g_create_mu);
  if (!ctx->lt) ctx->lt = new LtState;
  return *static_cast<LtState*>(ctx->lt);
}

#define LT_CHECK(call)                                                                                     \
  do {                                                                                                     \
    hipblasStatus_t st_ = (call);                                                                          \
    if (st_ != HIPBLAS_STATUS_SUCCESS) return qp_fail(QP_ERR_HIP, "qp_linear_act: %s failed (%d)", #call, (int)st_); \
  } while (0)

int handle_for(LtState& st, hipStream_t s, hipblasLtHandle_t* out) {
  static const bool shared = getenv("QP_LT_SHARED_HANDLE") != nullptr;     // developer switch: reproduce the stall (tests/concurrent_gemm_check.py)
  if (shared) s = nullptr;
  auto it = st.handles.find(s);
  if (it == st.handles.end()) {
    hipblasLtHandle_t h = nullptr;
    LT_CHECK(hipblasLtCreate(&h));
    it = st.handles.emplace(s, h).first;
  }
  *out = it->second;
  return QP_OK;
}

void destroy_plan(Plan& p) {
  if (p.desc) (void)hipblasLtMatmulDescDestroy(p.desc);
  if (p.a) (void)hipblasLtMatrixLayoutDestroy(p.a);
  if (p.b) (void)hipblasLtMatrixLayoutDestroy(p.b);
  if (p.d) (void)hipblasLtMatrixLayoutDestroy(p.d);
  p.desc = nullptr; p.a = p.b = p.d = nullptr;
}

int make_plan(int device, hipblasLtHandle_t handle, Plan& p, int64_t m, int64_t n, int64_t k, int act, int bias_kind, size_t max_ws) {
  const bool has_bias = bias_kind != 0;
  LT_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_DEFAULT;
  if (act == 0) epi = has_bias ? HIPBLASLT_EPILOGUE_BIAS : HIPBLASLT_EPILOGUE_DEFAULT;
  else epi = has_bias ? HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT : HIPBLASLT_EPILOGUE_SWISH_EXT;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
  if (has_bias) {
    const hipDataType bt = bias_kind == 2 ? HIP_R_32F : HIP_R_16BF;
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
  }
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.a, HIP_R_16BF, k, n, k));      // W: column-major [k x n], ld k
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.b, HIP_R_16BF, k, m, k));      // x: column-major [k x m], ld k
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.d, HIP_R_16BF, n, m, n));      // out: column-major [n x m], ld n
  hipblasLtMatmulPreference_t pref;
  LT_CHECK(hipblasLtMatmulPreferenceCreate(&pref));
  const uint64_t ws64 = max_ws;
  LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws64, sizeof(ws64)));
#ifndef QP_LT_MAX_CANDS
#define QP_LT_MAX_CANDS 32       // heuristic candidates the tuner times per problem (an A/B build with 128 found nothing faster: DESIGN 6)
#endif
  hipblasLtMatmulHeuristicResult_t res[QP_LT_MAX_CANDS];
  int found = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(handle, p.desc, p.a, p.b, p.d, p.d, pref, QP_LT_MAX_CANDS, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1)
    return qp_fail(QP_ERR_UNSUPPORTED, "qp_linear_act: hipBLASLt has no algorithm for m=%lld n=%lld k=%lld act=%d (status %d)", (long long)m,
                   (long long)n, (long long)k, act, (int)st);
  p.cands.assign(res, res + found);
  // the first candidate that fits the caller's workspace: this hipBLASLt build returns stream-K candidates that want 64 MB even when the
  // preference says 1 MB (found by the test of ADVICE r5's fix), so MAX_WORKSPACE_BYTES is a hint here, not a filter
  int pick = -1;
  for (int i = 0; i < found && pick < 0; ++i)
    if (res[i].workspaceSize <= max_ws) pick = i;
  if (pick < 0) {
    size_t need = res[0].workspaceSize;
    for (int i = 1; i < found; ++i) need = std::min(need, (size_t)res[i].workspaceSize);
    return qp_fail(QP_ERR_WORKSPACE, "qp_linear_act: workspace %zu < %zu bytes (the smallest any of hipBLASLt's %d candidates for m=%lld n=%lld k=%lld needs)",
                   max_ws, need, found, (long long)m, (long long)n, (long long)k);
  }
  Choice rec;                                                   // tuned earlier in this process (any context of this device): same algorithm
  if (recorded_choice(device, m, n, k, act, bias_kind, &rec)) {
    const int i = find_choice(p.cands, rec, max_ws);
    if (i >= 0) pick = i;
  }
#ifdef QP_EXPERIMENTS                                          // developer probe: QP_LT_ALGO_INDEX = i-th heuristic candidate
  if (const char* e = env_lt_algo_index()) { pick = atoi(e); if (pick >= found) pick = found - 1; if (pick < 0) pick = 0; }
#endif
  p.pick = pick;
  if (env_lt_debug()) fprintf(stderr, "[qp_linear_act] m=%lld n=%lld k=%lld act=%d: %d candidates, using %d (ws %zu)\n", (long long)m,
                                     (long long)n, (long long)k, act, found, pick, res[pick].workspaceSize);
  p.algo = res[pick].algo;
  p.ws = res[pick].workspaceSize;
  return QP_OK;
}

}  // namespace

void qp_lt_destroy(void* lt_state) {
  LtState* st = static_cast<LtState*>(lt_state);
  if (!st) return;
  for (auto& kv : st->plans) destroy_plan(kv.second);
  for (auto& kv : st->handles) (void)hipblasLtDestroy(kv.second);
  delete st;
}

int qp_launch_linear_act(qp_ctx* ctx, const void* x, const void* w, const void* bias, int bias_f32, float alpha, void* out, int64_t m, int64_t n,
                         int64_t k, int act, void* workspace, size_t workspace_bytes, hipStream_t s) {
  LtState& st = lt(ctx);
  std::lock_guard<std::mutex> g(st.mu);
  hipblasLtHandle_t handle = nullptr;
  if (int rc = handle_for(st, s, &handle)) return rc;
  const int bias_kind = bias ? (bias_f32 ? 2 : 1) : 0;
  const auto key = std::make_tuple(m, n, k, act, bias_kind);
  auto it = st.plans.find(key);
  if (it == st.plans.end()) {
    Plan p;
    int rc = make_plan(ctx->device, handle, p, m, n, k, act, bias_kind, workspace_bytes);
    if (rc) { destroy_plan(p); return rc; }
    it = st.plans.emplace(key, p).first;
  }
  Plan& p = it->second;
  {
    // another context of this device tuned the problem after this one planned it: converge on the recorded pick
    Choice rec;
    if (recorded_choice(ctx->device, m, n, k, act, bias_kind, &rec) && !same_algo(rec.algo, p.algo)) {
      const int i = find_choice(p.cands, rec, workspace_bytes);
      if (i >= 0) { p.algo = p.cands[i].algo; p.ws = p.cands[i].workspaceSize; p.pick = i; }
    }
  }
  if (p.ws > workspace_bytes) {
    // planned under a larger workspace than this call offers: the first candidate that fits (a different algorithm = possibly different
    // rounding; callers who want one set of bits keep one workspace size, as QuickPrefillOps does)
    int fit = -1;
    for (int i = 0; i < (int)p.cands.size() && fit < 0; ++i)
      if (p.cands[i].workspaceSize <= workspace_bytes) fit = i;
    if (fit < 0) return qp_fail(QP_ERR_WORKSPACE, "qp_linear_act: workspace %zu < %zu bytes", workspace_bytes, p.ws);
    p.algo = p.cands[fit].algo; p.ws = p.cands[fit].workspaceSize; p.pick = fit;
  }
  if (bias) LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  const float beta = 0.f;
  LT_CHECK(hipblasLtMatmul(handle, p.desc, &alpha, w, p.a, x, p.b, &beta, out, p.d, out, p.d, &p.algo, workspace, workspace_bytes, s));
  return QP_OK;
}

// Which heuristic candidate this context runs for the problem: *choice = its index in the context's own candidate list (0 =
// hipBLASLt's first pick), or -1 when the context has not seen the problem yet.  *tuned (optional) = 1 when the process holds a
// stopwatch decision for it on this device, 2 when this context's plan runs exactly that algorithm.
int qp_linear_plan_choice_impl(qp_ctx* ctx, int64_t m, int64_t n, int64_t k, int act, int bias_kind, int* choice, int* tuned) {
  LtState& st = lt(ctx);
  std::lock_guard<std::mutex> g(st.mu);
  Choice rec;
  const bool have = recorded_choice(ctx->device, m, n, k, act, bias_kind, &rec);
  auto it = st.plans.find(std::make_tuple(m, n, k, act, bias_kind));
  *choice = it == st.plans.end() ? -1 : it->second.pick;
  if (tuned) *tuned = !have ? 0 : (it != st.plans.end() && same_algo(rec.algo, it->second.algo) ? 2 : 1);
  return QP_OK;
}

// Times every heuristic candidate of this problem with COLD weights — the caller passes the same projection of several layers,
// visited round-robin, because a skinny GEMM re-reading one hot weight matrix is served by the Infinity Cache and ranks the
// candidates differently (prompt tail, M = 30, 7B dims: down projection 110 us with the default pick, 47 us with the best) —
// and keeps the fastest for later qp_linear_act calls of the same (m, n, k, act, bias kind).  Synchronises the stream.
int qp_launch_linear_tune(qp_ctx* ctx, const void* x, const void* const* ws_list, int n_ws, const void* bias, int bias_f32, float alpha, void* out,
                          int64_t m, int64_t n, int64_t k, int act, void* workspace, size_t workspace_bytes, hipStream_t s, int* chosen) {
  LtState& st = lt(ctx);
  std::lock_guard<std::mutex> g(st.mu);
  hipblasLtHandle_t handle = nullptr;
  if (int rc = handle_for(st, s, &handle)) return rc;
  const int bias_kind = bias ? (bias_f32 ? 2 : 1) : 0;
  const auto key = std::make_tuple(m, n, k, act, bias_kind);
  auto it = st.plans.find(key);
  if (it == st.plans.end()) {
    Plan p;
    int rc = make_plan(ctx->device, handle, p, m, n, k, act, bias_kind, workspace_bytes);
    if (rc) { destroy_plan(p); return rc; }
    it = st.plans.emplace(key, p).first;
  }
  Plan& p = it->second;
  {
    // already timed in this process on this device (by this or another context): adopt that pick, no second stopwatch run — two
    // runs of the tuner may disagree on candidates within timing noise of each other, and every context must compute the same bits
    // (a record this context's list does not hold — planned under another workspace limit — is re-timed here and replaced: the other
    // contexts converge on the new record at their next qp_linear_act)
    Choice rec;
    if (recorded_choice(ctx->device, m, n, k, act, bias_kind, &rec)) {
      const int i = find_choice(p.cands, rec, workspace_bytes);
      if (i >= 0) {
        p.algo = p.cands[i].algo; p.ws = p.cands[i].workspaceSize; p.pick = i;
        if (chosen) *chosen = i;
        return QP_OK;
      }
    }
  }
  if (bias) LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  const float beta = 0.f;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return qp_fail(QP_ERR_HIP, "qp_linear_tune: hipEventCreate");
  float best = 1e30f;
  int best_i = -1;
  std::vector<float> t_ms(p.cands.size(), -1.f);
  for (int c = 0; c < (int)p.cands.size(); ++c) {
    if (p.cands[c].workspaceSize > workspace_bytes) continue;
    bool ok = true;
    for (int rep = -2; rep < n_ws && ok; ++rep) {               // two untimed runs, then one pass over the weight list
      if (rep == 0) (void)hipEventRecord(e0, s);
      const void* w = ws_list[(rep + 2 * n_ws) % n_ws];
      ok = hipblasLtMatmul(handle, p.desc, &alpha, w, p.a, x, p.b, &beta, out, p.d, out, p.d, &p.cands[c].algo, workspace, workspace_bytes,
                           s) == HIPBLAS_STATUS_SUCCESS;
    }
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) ok = false;
    float ms = 0.f;
    if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { t_ms[c] = ms; if (ms < best) { best = ms; best_i = c; } }
  }
  // Two candidates within timing noise of each other would otherwise win on alternate runs, and a different algorithm is a different
  // fp32 accumulation order (after 450 groups x 28 layers the first generated token of the 1-hour benchmark flipped between two values
  // from run to run).  Take the LOWEST-INDEX candidate within 2 % of the fastest: the choice only moves when a candidate sits on that
  // boundary, at a cost of at most 2 % on one projection.
  for (int c = 0; c < best_i; ++c)
    if (t_ms[c] > 0.f && t_ms[c] <= 1.02f * best) { best_i = c; break; }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  // Every candidate has just used the caller's workspace its own way; stream-K kernels keep their arrival flags there and expect to find
  // them zero (they reset them on the way out).  Leave the workspace as a fresh allocation would be, so that the algorithm picked here
  // never meets another candidate's partial tiles where it expects its flags.
  if (workspace && workspace_bytes && (hipMemsetAsync(workspace, 0, workspace_bytes, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
    return qp_fail(QP_ERR_HIP, "qp_linear_tune: clearing the workspace failed");
  if (best_i < 0) return qp_fail(QP_ERR_HIP, "qp_linear_tune: no candidate ran");
  p.algo = p.cands[best_i].algo;
  p.ws = p.cands[best_i].workspaceSize;
  p.pick = best_i;
  {
    std::lock_guard<std::mutex> gc(g_choice_mu);
    Choice c;
    c.algo = p.algo; c.ws = p.ws;
    g_choice[std::make_tuple(ctx->device, m, n, k, act, bias_kind)] = c;
  }
  if (chosen) *chosen = best_i;
  if (env_lt_debug()) fprintf(stderr, "[qp_linear_tune] m=%lld n=%lld k=%lld: candidate %d of %zu, %.1f us per call\n", (long long)m,
                                     (long long)n, (long long)k, best_i, p.cands.size(), best * 1e3f / n_ws);
  return QP_OK;
}
