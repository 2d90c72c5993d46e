// C ABI of libquickprefill.so (include/quickprefill.h): argument validation + launches.  No torch types,
// no device allocation, no synchronisation: everything is enqueued on the caller's stream.
#include "qp_common.h"
#include <mutex>
#include <string>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";

int qp_fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

int qp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return qp_fail(QP_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
  return QP_OK;
}

qp_dev_switches& qp_dev() {
  static qp_dev_switches* sw = [] {
    qp_dev_switches* d = new qp_dev_switches;
    auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    d->attn_variant = geti("QP_ATTN_VARIANT", 0);
    d->attn_force_split = geti("QP_ATTN_FORCE_SPLIT", 0);
    d->s6_prio = geti("QP_S6_PRIO", 0) & 3;
    d->s6_early_out = geti("QP_S6_EARLY_OUT", 3) & 3;
    { const char* e = getenv("QP_DECODE_ATTN"); d->decode_attn_valu = (e && e[0] == 'v') ? 1 : 0; }
    d->attn_debug = getenv("QP_ATTN_DEBUG") != nullptr ? 1 : 0;
    d->attn_flat = geti("QP_ATTN_FLAT", -1);
    return d;
  }();
  return *sw;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" {

const char* qp_last_error(void) { return g_err; }
const char* qp_version(void) { return "quickprefill-mi355x 0.7 (gfx950)"; }

int qp_dev_switch(const char* name, int value) {
  QP_REQUIRE(name != nullptr, QP_ERR_INVALID, "qp_dev_switch: name is NULL");
  qp_dev_switches& d = qp_dev();
  const std::string k(name);
  if (k == "attn_variant") d.attn_variant = value;
  else if (k == "attn_force_split") d.attn_force_split = value;
  else if (k == "s6_prio") d.s6_prio = value & 3;
  else if (k == "s6_early_out") d.s6_early_out = value & 3;
  else if (k == "decode_attn_valu") d.decode_attn_valu = value ? 1 : 0;
  else if (k == "attn_debug") d.attn_debug = value ? 1 : 0;
  else if (k == "attn_flat") d.attn_flat = value < 0 ? -1 : (value ? 1 : 0);
  else return qp_fail(QP_ERR_INVALID, "qp_dev_switch: unknown switch '%s'", name);
  return QP_OK;
}

int qp_create(qp_ctx** out, int device) {
  QP_REQUIRE(out != nullptr, QP_ERR_INVALID, "qp_create: out is NULL");
  (void)qp_dev();                                          // developer switches: environment read once, here
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return qp_fail(QP_ERR_HIP, "qp_create: no HIP device (%s)", hipGetErrorString(e));
  QP_REQUIRE(device >= 0 && device < count, QP_ERR_INVALID, "qp_create: device %d out of range [0,%d)", device, count);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return qp_fail(QP_ERR_HIP, "qp_create: hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return qp_fail(QP_ERR_UNSUPPORTED, "qp_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                   prop.gcnArchName);
  qp_ctx* c = new (std::nothrow) qp_ctx;
  QP_REQUIRE(c != nullptr, QP_ERR_INVALID, "qp_create: out of host memory");
  c->device = device;
  c->cus = prop.multiProcessorCount;
  c->lds_per_cu = (int)prop.sharedMemPerBlock;
  c->lt = nullptr;
  *out = c;
  return QP_OK;
}

void qp_destroy(qp_ctx* ctx) {
  if (!ctx) return;
  qp_lt_destroy(ctx->lt);                      // hipBLASLt handles / plans of this context
  delete ctx;
}

static inline bool valid_mode(int m) { return m >= 0 && m <= 3; }

// Host-side, GIL-free ring fill of the overlap producer (qwen25_lvu_interleaved.py:303-340 runs `.float()` + the HF processor under
// the GIL): a plain memcpy into the pinned slot, split over a few std::threads for large groups.  Called through ctypes, which
// drops the GIL for the duration of the call.  (Measured in the build container, 100 MB blocks: 4.8 ms with 4 threads vs 8.9 ms
// for torch's Tensor.copy_ with 8 intra-op threads and 53 ms with 1 — torch also drops the GIL, main-thread stalls stayed
// <= 7 ms, but its speed depends on the process-wide intra-op thread setting, which the CPU baseline and user code change.)
int qp_host_memcpy(void* dst, const void* src, size_t bytes, int threads) {
  QP_REQUIRE(dst && src, QP_ERR_INVALID, "qp_host_memcpy: NULL argument");
  if (bytes == 0) return QP_OK;
  if (threads < 1) threads = 1;
  if (threads > 16) threads = 16;
  const size_t min_chunk = 4u << 20;
  if ((size_t)threads > bytes / min_chunk) threads = (int)(bytes / min_chunk);
  if (threads <= 1) { memcpy(dst, src, bytes); return QP_OK; }
  std::vector<std::thread> pool;
  const size_t chunk = ((bytes / threads) + 4095) & ~(size_t)4095;
  for (int t = 1; t < threads; ++t) {
    const size_t off = chunk * t;
    if (off >= bytes) break;
    const size_t len = off + chunk > bytes ? bytes - off : chunk;
    pool.emplace_back([=] { memcpy((char*)dst + off, (const char*)src + off, len); });
  }
  memcpy(dst, src, chunk < bytes ? chunk : bytes);
  for (auto& th : pool) th.join();
  return QP_OK;
}
int qp_device_cus(const qp_ctx* ctx) { return ctx ? ctx->cus : 0; }

int qp_mrope_table(qp_ctx* ctx, const int64_t* pos, int64_t n, const int32_t sections[3], float theta, int head_dim,
                   void* cos_out, void* sin_out, void* stream) {
  QP_REQUIRE(ctx && pos && sections && cos_out && sin_out, QP_ERR_INVALID, "qp_mrope_table: NULL argument");
  QP_REQUIRE(n >= 0, QP_ERR_INVALID, "qp_mrope_table: n=%lld", (long long)n);
  QP_REQUIRE(head_dim > 0 && head_dim % 2 == 0 && sections[0] + sections[1] + sections[2] == head_dim / 2, QP_ERR_INVALID,
             "qp_mrope_table: mrope sections %d+%d+%d must sum to head_dim/2=%d", sections[0], sections[1], sections[2],
             head_dim / 2);
  if (n == 0) return QP_OK;
  return qp_launch_mrope_table(pos, n, sections, theta, head_dim, cos_out, sin_out, (hipStream_t)stream);
}

int qp_rope_append(qp_ctx* ctx, const void* qkv, const void* cos, const void* sin, int64_t n, int n_q_heads, int n_kv_heads,
                   int head_dim, void* q_out, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0,
                   float* head_sumsq, void* stream) {
  QP_REQUIRE(ctx && qkv && cos && sin && q_out && k_dst && v_dst, QP_ERR_INVALID, "qp_rope_append: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_rope_append: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n >= 0 && n_q_heads > 0 && n_kv_heads > 0 && dst_row0 >= 0, QP_ERR_INVALID, "qp_rope_append: bad sizes");
  QP_REQUIRE(dst_head_stride % 8 == 0 && dst_head_stride >= (dst_row0 + n) * head_dim, QP_ERR_INVALID,
             "qp_rope_append: head stride %lld too small for rows [%lld,%lld)", (long long)dst_head_stride, (long long)dst_row0,
             (long long)(dst_row0 + n));
  QP_REQUIRE(aligned16(qkv) && aligned16(cos) && aligned16(sin) && aligned16(q_out) && aligned16(k_dst) && aligned16(v_dst),
             QP_ERR_INVALID, "qp_rope_append: pointers must be 16-byte aligned");
  if (n == 0) return QP_OK;
  return qp_launch_rope_append(qkv, cos, sin, n, n_q_heads, n_kv_heads, q_out, k_dst, v_dst, dst_head_stride, dst_row0,
                               head_sumsq, nullptr, 0, (hipStream_t)stream);
}

int qp_rope_append_keys(qp_ctx* ctx, const void* qkv, const void* cos, const void* sin, int64_t n, int n_q_heads, int n_kv_heads,
                        int head_dim, void* q_out, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0,
                        float* head_sumsq, uint16_t* norm_keys, int prune_mode, void* stream) {
  QP_REQUIRE(ctx && qkv && cos && sin && q_out && k_dst && v_dst && norm_keys, QP_ERR_INVALID, "qp_rope_append_keys: NULL argument");
  QP_REQUIRE(prune_mode == QP_PRUNE_KEY_NORMS_SMALL || prune_mode == QP_PRUNE_KEY_NORMS, QP_ERR_INVALID,
             "qp_rope_append_keys: prune_mode=%d (this entry point scores the KEY rows: QP_PRUNE_KEY_NORMS_SMALL or QP_PRUNE_KEY_NORMS)", prune_mode);
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_rope_append_keys: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n >= 0 && n_q_heads > 0 && n_kv_heads > 0 && dst_row0 >= 0, QP_ERR_INVALID, "qp_rope_append_keys: bad sizes");
  QP_REQUIRE(qp_rope_can_fuse_keys(n_q_heads, n_kv_heads), QP_ERR_UNSUPPORTED,
             "qp_rope_append_keys: %d q / %d kv heads: a token's key rows do not share a wave (use qp_rope_append + qp_norm_keys)",
             n_q_heads, n_kv_heads);
  QP_REQUIRE(dst_head_stride % 8 == 0 && dst_head_stride >= (dst_row0 + n) * head_dim, QP_ERR_INVALID,
             "qp_rope_append_keys: head stride %lld too small for rows [%lld,%lld)", (long long)dst_head_stride, (long long)dst_row0,
             (long long)(dst_row0 + n));
  QP_REQUIRE(aligned16(qkv) && aligned16(cos) && aligned16(sin) && aligned16(q_out) && aligned16(k_dst) && aligned16(v_dst),
             QP_ERR_INVALID, "qp_rope_append_keys: pointers must be 16-byte aligned");
  if (n == 0) return QP_OK;
  return qp_launch_rope_append(qkv, cos, sin, n, n_q_heads, n_kv_heads, q_out, k_dst, v_dst, dst_head_stride, dst_row0,
                               head_sumsq, norm_keys, qp_mode_largest(prune_mode), (hipStream_t)stream);
}

size_t qp_query_scores_workspace_bytes(int64_t n, int64_t m, int n_q_heads) {
  if (n <= 0 || m <= 0 || n_q_heads <= 0) return 0;
  return (((size_t)n_q_heads * m * n * 2 + 255) & ~(size_t)255) + (size_t)n_q_heads * n * 2 + 256;      // probabilities + per-head sums
}

int qp_query_head_sums(qp_ctx* ctx, const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int n_q_heads,
                       int n_kv_heads, int head_dim, uint16_t* head_sums_out, void* workspace, size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && q_prompt && k_group && head_sums_out && workspace, QP_ERR_INVALID, "qp_query_head_sums: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_query_head_sums: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n > 0 && m > 0 && n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, QP_ERR_INVALID, "qp_query_head_sums: bad sizes");
  QP_REQUIRE(n <= 32768 && m <= 65535 && n_q_heads <= 65535, QP_ERR_UNSUPPORTED, "qp_query_head_sums: n=%lld > 32768 (one softmax row lives in LDS)",
             (long long)n);
  QP_REQUIRE(k_head_stride % 8 == 0 && k_head_stride >= n * head_dim, QP_ERR_INVALID, "qp_query_head_sums: bad key stride");
  QP_REQUIRE(workspace_bytes >= qp_query_scores_workspace_bytes(n, m, n_q_heads), QP_ERR_WORKSPACE, "qp_query_head_sums: workspace %zu < %zu bytes",
             workspace_bytes, qp_query_scores_workspace_bytes(n, m, n_q_heads));
  QP_REQUIRE(aligned16(q_prompt) && aligned16(k_group) && aligned16(workspace), QP_ERR_INVALID, "qp_query_head_sums: alignment");
  return qp_launch_query_head_sums(q_prompt, k_group, k_head_stride, n, m, n_q_heads, n_kv_heads, head_sums_out, workspace, (hipStream_t)stream);
}

int qp_query_scores_from_head_sums(qp_ctx* ctx, const uint16_t* head_sums, int n_heads_total, int64_t n, const float* value_sumsq,
                                   int n_kv_heads_total, uint16_t* norm_keys_out, uint16_t* scores_out, void* stream) {
  QP_REQUIRE(ctx && head_sums && norm_keys_out, QP_ERR_INVALID, "qp_query_scores_from_head_sums: NULL argument");
  QP_REQUIRE(n_heads_total > 0 && n > 0 && n < (1ll << 31) && (!value_sumsq || n_kv_heads_total > 0), QP_ERR_INVALID,
             "qp_query_scores_from_head_sums: bad sizes");
  return qp_launch_query_scores_final(head_sums, n_heads_total, n, value_sumsq, n_kv_heads_total, norm_keys_out, scores_out, (hipStream_t)stream);
}

int qp_query_scores(qp_ctx* ctx, const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int n_q_heads,
                    int n_kv_heads, int head_dim, const float* value_sumsq, uint16_t* norm_keys_out, uint16_t* scores_out, void* workspace,
                    size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && q_prompt && k_group && norm_keys_out && workspace, QP_ERR_INVALID, "qp_query_scores: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_query_scores: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n > 0 && m > 0 && n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, QP_ERR_INVALID, "qp_query_scores: bad sizes");
  QP_REQUIRE(n <= 32768 && m <= 65535 && n_q_heads <= 65535, QP_ERR_UNSUPPORTED, "qp_query_scores: n=%lld > 32768 (one softmax row lives in LDS)",
             (long long)n);
  QP_REQUIRE(k_head_stride % 8 == 0 && k_head_stride >= n * head_dim, QP_ERR_INVALID, "qp_query_scores: bad key stride");
  QP_REQUIRE(workspace_bytes >= qp_query_scores_workspace_bytes(n, m, n_q_heads), QP_ERR_WORKSPACE, "qp_query_scores: workspace %zu < %zu bytes",
             workspace_bytes, qp_query_scores_workspace_bytes(n, m, n_q_heads));
  QP_REQUIRE(aligned16(q_prompt) && aligned16(k_group) && aligned16(workspace), QP_ERR_INVALID, "qp_query_scores: alignment");
  return qp_launch_query_scores(q_prompt, k_group, k_head_stride, n, m, n_q_heads, n_kv_heads, value_sumsq, norm_keys_out, scores_out,
                                workspace, (hipStream_t)stream);
}

int qp_norm_keys(qp_ctx* ctx, const float* head_sumsq, int n_heads_total, int64_t n, uint16_t* norm_keys, int prune_mode, void* stream) {
  QP_REQUIRE(ctx && head_sumsq && norm_keys, QP_ERR_INVALID, "qp_norm_keys: NULL argument");
  QP_REQUIRE(valid_mode(prune_mode), QP_ERR_INVALID, "qp_norm_keys: prune_mode=%d (0..3, enum qp_prune_mode)", prune_mode);
  QP_REQUIRE(n_heads_total > 0 && n > 0 && n < (1ll << 31), QP_ERR_INVALID, "qp_norm_keys: need n_heads_total > 0 and n > 0 (n=%lld)", (long long)n);
  return qp_launch_norm_keys(head_sumsq, n_heads_total, n, norm_keys, qp_mode_largest(prune_mode), (hipStream_t)stream);
}

int qp_prune_keys(qp_ctx* ctx, const uint16_t* norm_keys, int64_t n, int64_t k, const void* k_src, const void* v_src,
                  int64_t src_head_stride, int n_kv_heads, int head_dim, void* k_dst, void* v_dst, int64_t dst_head_stride,
                  int64_t dst_row0, int32_t* kept_idx_out, void* stream) {
  QP_REQUIRE(ctx && norm_keys && k_src && v_src && k_dst && v_dst && kept_idx_out, QP_ERR_INVALID, "qp_prune_keys: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_prune_keys: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(k > 0 && k <= n && n_kv_heads > 0 && dst_row0 >= 0, QP_ERR_INVALID, "qp_prune_keys: need 0 < k <= n, got k=%lld n=%lld",
             (long long)k, (long long)n);
  QP_REQUIRE(n <= 8192, QP_ERR_UNSUPPORTED, "qp_prune_keys: n=%lld > 8192 (use qp_select_k_smallest + qp_gather_kv)", (long long)n);
  QP_REQUIRE(n_kv_heads <= 8, QP_ERR_UNSUPPORTED, "qp_prune_keys: n_kv_heads=%d > 8", n_kv_heads);
  QP_REQUIRE(src_head_stride % 8 == 0 && dst_head_stride % 8 == 0 && src_head_stride >= n * head_dim && dst_head_stride >= (dst_row0 + k) * head_dim,
             QP_ERR_INVALID, "qp_prune_keys: bad strides");
  QP_REQUIRE(aligned16(k_src) && aligned16(v_src) && aligned16(k_dst) && aligned16(v_dst) && aligned16(norm_keys), QP_ERR_INVALID,
             "qp_prune_keys: pointers must be 16-byte aligned");
  return qp_launch_prune_keys(norm_keys, n, k, k_src, v_src, src_head_stride, n_kv_heads, k_dst, v_dst, dst_head_stride, dst_row0,
                              kept_idx_out, (hipStream_t)stream);
}

int qp_prefill_attn_rows(qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix, int64_t prefix_head_stride,
                         int64_t prefix_len, const void* k_new, const void* v_new, int64_t new_head_stride, int64_t n, int64_t q_row0,
                         int64_t nq, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* out, void* workspace,
                         size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && q && k_new && v_new && out, QP_ERR_INVALID, "qp_prefill_attn: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_prefill_attn: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n >= 0 && prefix_len >= 0, QP_ERR_INVALID, "qp_prefill_attn: negative length");
  QP_REQUIRE(q_row0 >= 0 && nq >= 0 && q_row0 + nq <= n, QP_ERR_INVALID, "qp_prefill_attn: query rows [%lld,%lld) outside the group's %lld tokens",
             (long long)q_row0, (long long)(q_row0 + nq), (long long)n);
  QP_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, QP_ERR_INVALID,
             "qp_prefill_attn: n_q_heads=%d not a multiple of n_kv_heads=%d", n_q_heads, n_kv_heads);
  QP_REQUIRE(prefix_len == 0 || (k_prefix && v_prefix && prefix_head_stride >= prefix_len * head_dim), QP_ERR_INVALID,
             "qp_prefill_attn: prefix pointers/stride invalid for prefix_len=%lld", (long long)prefix_len);
  QP_REQUIRE(new_head_stride >= n * head_dim && new_head_stride % 8 == 0 && prefix_head_stride % 8 == 0, QP_ERR_INVALID,
             "qp_prefill_attn: bad head strides");
  QP_REQUIRE(aligned16(q) && aligned16(k_new) && aligned16(v_new) && aligned16(out) && aligned16(k_prefix) && aligned16(v_prefix),
             QP_ERR_INVALID, "qp_prefill_attn: pointers must be 16-byte aligned");
  QP_REQUIRE(n_q_heads <= 65535, QP_ERR_UNSUPPORTED, "qp_prefill_attn: too many heads");
  QP_REQUIRE(workspace == nullptr || aligned16(workspace), QP_ERR_INVALID, "qp_prefill_attn: workspace must be 16-byte aligned");
  if (nq == 0) return QP_OK;
  return qp_launch_prefill_attn(ctx, q, k_prefix, v_prefix, prefix_head_stride, prefix_len, k_new, v_new, new_head_stride, n, q_row0,
                                nq, n_q_heads, n_kv_heads, scale, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int qp_prefill_attn(qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix, int64_t prefix_head_stride,
                    int64_t prefix_len, const void* k_new, const void* v_new, int64_t new_head_stride, int64_t n,
                    int n_q_heads, int n_kv_heads, int head_dim, float scale, void* out, void* workspace, size_t workspace_bytes,
                    void* stream) {
  return qp_prefill_attn_rows(ctx, q, k_prefix, v_prefix, prefix_head_stride, prefix_len, k_new, v_new, new_head_stride, n, 0, n,
                              n_q_heads, n_kv_heads, head_dim, scale, out, workspace, workspace_bytes, stream);
}

size_t qp_attn_workspace_bytes(const qp_ctx* ctx, int64_t n, int64_t prefix_len, int n_q_heads, int n_kv_heads) {
  if (!ctx || n <= 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_q_heads % n_kv_heads) return 0;
  return qp_attn_workspace_bytes_impl(ctx, n, prefix_len, n_q_heads, n_kv_heads);
}

int qp_key_sumsq(qp_ctx* ctx, const void* k, int64_t head_stride, int64_t row0, int64_t n, int n_kv_heads, int head_dim,
                 float* head_sumsq, void* stream) {
  QP_REQUIRE(ctx && k && head_sumsq, QP_ERR_INVALID, "qp_key_sumsq: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_key_sumsq: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n >= 0 && row0 >= 0 && n_kv_heads > 0 && head_stride % 8 == 0 && head_stride >= (row0 + n) * head_dim,
             QP_ERR_INVALID, "qp_key_sumsq: bad sizes");
  QP_REQUIRE(aligned16(k), QP_ERR_INVALID, "qp_key_sumsq: k must be 16-byte aligned");
  if (n == 0) return QP_OK;
  return qp_launch_key_sumsq(k, head_stride, row0, n, n_kv_heads, head_sumsq, (hipStream_t)stream);
}

size_t qp_select_workspace_bytes(int64_t n) { return n > 65536 ? (size_t)n * 2 + 256 : 256; }

int qp_select_k_smallest(qp_ctx* ctx, const float* head_sumsq, int n_heads_total, int64_t n, int64_t k,
                         int32_t* kept_idx_out, uint16_t* norm_bits_out, int prune_mode, void* workspace, size_t workspace_bytes,
                         void* stream) {
  QP_REQUIRE(ctx && head_sumsq && kept_idx_out, QP_ERR_INVALID, "qp_select_k_smallest: NULL argument");
  QP_REQUIRE(valid_mode(prune_mode), QP_ERR_INVALID, "qp_select_k_smallest: prune_mode=%d (0..3, enum qp_prune_mode)", prune_mode);
  QP_REQUIRE(n_heads_total > 0, QP_ERR_INVALID, "qp_select_k_smallest: n_heads_total=%d", n_heads_total);
  QP_REQUIRE(k > 0 && k <= n, QP_ERR_INVALID, "qp_select_k_smallest: need 0 < k <= n, got k=%lld n=%lld", (long long)k,
             (long long)n);
  QP_REQUIRE(n < (1ll << 31), QP_ERR_UNSUPPORTED, "qp_select_k_smallest: n=%lld too large", (long long)n);
  QP_REQUIRE(n <= 65536 || (workspace != nullptr && workspace_bytes >= qp_select_workspace_bytes(n)), QP_ERR_WORKSPACE,
             "qp_select_k_smallest: n=%lld > 65536 needs a workspace of %zu bytes", (long long)n, qp_select_workspace_bytes(n));
  return qp_launch_select(head_sumsq, n_heads_total, n, k, kept_idx_out, norm_bits_out, workspace, qp_mode_largest(prune_mode),
                          (hipStream_t)stream);
}

int qp_select_keys(qp_ctx* ctx, const uint16_t* norm_keys, int64_t n, int64_t k, int32_t* kept_idx_out, void* stream) {
  QP_REQUIRE(ctx && norm_keys && kept_idx_out, QP_ERR_INVALID, "qp_select_keys: NULL argument");
  QP_REQUIRE(k > 0 && k <= n, QP_ERR_INVALID, "qp_select_keys: need 0 < k <= n, got k=%lld n=%lld", (long long)k, (long long)n);
  QP_REQUIRE(n < (1ll << 31), QP_ERR_UNSUPPORTED, "qp_select_keys: n=%lld too large", (long long)n);
  return qp_launch_select(nullptr, 0, n, k, kept_idx_out, nullptr, nullptr, 0, (hipStream_t)stream, norm_keys);
}

int qp_gather_kv(qp_ctx* ctx, const void* k_src, const void* v_src, int64_t src_head_stride, const int32_t* idx, int64_t k,
                 int n_kv_heads, int head_dim, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0,
                 void* stream) {
  QP_REQUIRE(ctx && k_src && v_src && idx && k_dst && v_dst, QP_ERR_INVALID, "qp_gather_kv: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_gather_kv: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(k >= 0 && n_kv_heads > 0 && dst_row0 >= 0 && src_head_stride % 8 == 0 && dst_head_stride % 8 == 0 &&
                 dst_head_stride >= (dst_row0 + k) * head_dim,
             QP_ERR_INVALID, "qp_gather_kv: bad sizes");
  QP_REQUIRE(aligned16(k_src) && aligned16(v_src) && aligned16(k_dst) && aligned16(v_dst), QP_ERR_INVALID,
             "qp_gather_kv: pointers must be 16-byte aligned");
  if (k == 0) return QP_OK;
  return qp_launch_gather_kv(k_src, v_src, src_head_stride, idx, k, n_kv_heads, k_dst, v_dst, dst_head_stride, dst_row0,
                             (hipStream_t)stream);
}

int qp_prune_staged(qp_ctx* ctx, const float* head_sumsq, int n_heads_total, int64_t n, int64_t k, const void* k_src,
                    const void* v_src, int64_t src_head_stride, int n_kv_heads, int head_dim, void* k_dst, void* v_dst,
                    int64_t dst_head_stride, int64_t dst_row0, int32_t* kept_idx_out, uint16_t* norm_bits_out, int prune_mode,
                    void* stream) {
  QP_REQUIRE(ctx && head_sumsq && k_src && v_src && k_dst && v_dst && kept_idx_out, QP_ERR_INVALID, "qp_prune_staged: NULL argument");
  QP_REQUIRE(valid_mode(prune_mode), QP_ERR_INVALID, "qp_prune_staged: prune_mode=%d (0..3, enum qp_prune_mode)", prune_mode);
  const int largest = qp_mode_largest(prune_mode);
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_prune_staged: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n_heads_total > 0 && n_kv_heads > 0 && k > 0 && k <= n, QP_ERR_INVALID,
             "qp_prune_staged: need 0 < k <= n, got k=%lld n=%lld", (long long)k, (long long)n);
  QP_REQUIRE(n <= 65536, QP_ERR_UNSUPPORTED, "qp_prune_staged: n=%lld > 65536", (long long)n);
  QP_REQUIRE(dst_row0 >= 0 && src_head_stride % 8 == 0 && dst_head_stride % 8 == 0 && dst_head_stride >= (dst_row0 + k) * head_dim,
             QP_ERR_INVALID, "qp_prune_staged: bad strides");
  QP_REQUIRE(aligned16(k_src) && aligned16(v_src) && aligned16(k_dst) && aligned16(v_dst), QP_ERR_INVALID, "qp_prune_staged: alignment");
  hipStream_t s = (hipStream_t)stream;
  // two launches (round 1's fused form — every one of up to 256 workgroups repeating the whole select in 150 KB of LDS — is gone: the
  // engine prunes through qp_prune_keys, and this entry point only serves groups of n > 8192 tokens, where a launch boundary is noise)
  int rc = qp_launch_select(head_sumsq, n_heads_total, n, k, kept_idx_out, norm_bits_out, nullptr, largest, s);
  if (rc) return rc;
  return qp_launch_gather_kv(k_src, v_src, src_head_stride, kept_idx_out, k, n_kv_heads, k_dst, v_dst, dst_head_stride, dst_row0, s);
}

// qp_prune_tail.  n <= 8192 and n_kv_heads <= 8 (every group size the reference is run at): TWO launches, no HBM bounce —
//   workspace = [norm keys uint16 n | pad to 256][one int32 "rows loaded" flag per 16-token slice]
// Larger groups (the single-group baseline mode of long videos) keep the round-1 form: sums -> one-workgroup select -> gather into
// the workspace -> copy back;  workspace = [head_sumsq fp32 Hkv*n | pad][K rows Hkv*k*D bf16 | pad][V rows | pad].
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline bool tail_inplace_ok(int64_t n, int n_kv_heads) {
#ifdef QP_EXPERIMENTS
  static const bool staged = getenv("QP_PRUNE_TAIL_STAGED") != nullptr;    // `make EXPERIMENTS=1` only: A/B switch to the round-1 form (tools/bench_prune_tail.py)
  if (staged) return false;
#endif
  return n <= 8192 && n_kv_heads <= 8;
}
static inline int64_t tail_slices(int64_t n) { return (n + 15) / 16; }

static size_t staged_tail_bytes(int64_t n, int64_t k, int n_kv_heads, int head_dim) {
  return align256((size_t)n_kv_heads * n * 4) + 2 * align256((size_t)n_kv_heads * k * head_dim * 2) + 256;
}
static size_t inplace_tail_bytes(int64_t n) { return align256((size_t)n * 2) + align256((size_t)tail_slices(n) * 4); }

// qp_prune_tail's in-place launch spin-waits on lower-numbered workgroups, which is only deadlock-free while its WHOLE grid can be
// resident.  At most ONE such grid is in flight per device, whichever context or stream launches it: the event of the last launch lives
// here (never destroyed: a handful of bytes per device for the life of the process).
struct TailGuard { std::mutex mu; hipEvent_t ev = nullptr; hipStream_t stream = nullptr; bool pending = false; };
static TailGuard* tail_guard(int device) {
  static TailGuard table[64];
  return &table[device & 63];
}

// CUs `s` may run on: the population count of its CU mask (hipExtStreamCreateWithCUMask / HSA_CU_MASK); the device's count when the
// runtime reports none.  The in-place grid must fit on THESE, not on the whole device.
static int qp_stream_cus(const qp_ctx* ctx, hipStream_t s) {
  uint32_t mask[32] = {0};
  if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) { (void)hipGetLastError(); return ctx->cus; }
  int bits = 0;
  for (int i = 0; i < 32; ++i) bits += __builtin_popcount(mask[i]);
  return (bits > 0 && bits < ctx->cus) ? bits : ctx->cus;
}

// Upper bound that is sufficient whatever form qp_prune_tail picks (no context: the device's capacity is unknown here).
size_t qp_prune_workspace_bytes(int64_t n, int64_t k, int n_kv_heads, int head_dim) {
  if (n <= 0 || k <= 0 || n_kv_heads <= 0 || head_dim <= 0) return 256;
  const size_t a = staged_tail_bytes(n, k, n_kv_heads, head_dim), b = inplace_tail_bytes(n);
  return a > b ? a : b;
}

// Exact size for THIS context on `stream` (NULL = an unmasked stream): the small in-place figure (2.25 B per token) only when
// qp_prune_tail will take the in-place form there — the same decision, from the same inputs.
size_t qp_prune_tail_workspace_bytes(const qp_ctx* ctx, int64_t n, int64_t k, int n_kv_heads, int head_dim, void* stream) {
  if (n <= 0 || k <= 0 || n_kv_heads <= 0 || head_dim <= 0) return 256;
  if (ctx && tail_inplace_ok(n, n_kv_heads) &&
      tail_slices(n) <= qp_prune_tail_inplace_capacity(qp_stream_cus(ctx, (hipStream_t)stream)))   // the SAME query the launcher makes, NULL stream
                                                                                                // included (a global HSA_CU_MASK masks it too)
    return inplace_tail_bytes(n);
  return staged_tail_bytes(n, k, n_kv_heads, head_dim);
}

int qp_prune_tail(qp_ctx* ctx, void* k_cache, void* v_cache, int64_t head_stride, int64_t past_len, int64_t n, int64_t k,
                  int n_kv_heads, int head_dim, int32_t* kept_idx_out, int prune_mode, void* workspace, size_t workspace_bytes,
                  void* stream) {
  QP_REQUIRE(ctx && k_cache && v_cache && kept_idx_out && workspace, QP_ERR_INVALID, "qp_prune_tail: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_prune_tail: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(valid_mode(prune_mode), QP_ERR_INVALID, "qp_prune_tail: prune_mode=%d (0..3, enum qp_prune_mode)", prune_mode);
  QP_REQUIRE(past_len >= 0 && k > 0 && k <= n && n_kv_heads > 0, QP_ERR_INVALID,
             "qp_prune_tail: need past_len >= 0 and 0 < k <= n (k=%lld n=%lld)", (long long)k, (long long)n);
  QP_REQUIRE(n <= 65536, QP_ERR_UNSUPPORTED, "qp_prune_tail: n=%lld > 65536", (long long)n);
  QP_REQUIRE(head_stride % 8 == 0 && head_stride >= (past_len + n) * head_dim, QP_ERR_INVALID, "qp_prune_tail: head stride too small");
  hipStream_t s = (hipStream_t)stream;
  // in place only when the whole grid of the second launch can be resident at once on the CUs this stream may use (see qp_prune.hip:
  // that, and one such grid in flight per context, is what makes its slice-ordered wait deadlock-free); 512 workgroups at most, an
  // unmasked MI355X holds 1024
  const bool inplace = tail_inplace_ok(n, n_kv_heads) && tail_slices(n) <= qp_prune_tail_inplace_capacity(qp_stream_cus(ctx, s));
  const size_t need = inplace ? inplace_tail_bytes(n) : staged_tail_bytes(n, k, n_kv_heads, head_dim);
  QP_REQUIRE(workspace_bytes >= need, QP_ERR_WORKSPACE, "qp_prune_tail: workspace %zu < %zu bytes%s", workspace_bytes, need,
             (!inplace && tail_inplace_ok(n, n_kv_heads)) ? " (this stream's CUs cannot hold the in-place grid at once: the staged form needs the larger "
             "scratch — size it with qp_prune_workspace_bytes, or qp_prune_tail_workspace_bytes for this stream)" : "");
  QP_REQUIRE(aligned16(k_cache) && aligned16(v_cache) && aligned16(workspace), QP_ERR_INVALID, "qp_prune_tail: alignment");
  unsigned char* ws = (unsigned char*)workspace;
  const int largest = qp_mode_largest(prune_mode);
  const void* scored = qp_mode_values(prune_mode) ? v_cache : k_cache;
  if (inplace) {
    uint16_t* keys = (uint16_t*)ws;
    int* sync_words = (int*)(ws + align256((size_t)n * 2));
    // ONE in-place grid in flight per DEVICE (round 5: the guard lives in a process-wide table indexed by the device, not in the
    // context — three contexts on one device could otherwise oversubscribe the CUs again): a call on another stream than the previous
    // one waits (on the device) for that one's event.  Inside a stream capture the event dance is skipped — the captured stream orders
    // its own nodes; a caller that replays several such graphs at once owns that ordering.
    TailGuard* tg = tail_guard(ctx->device);
    std::lock_guard<std::mutex> guard(tg->mu);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
    if (!capturing) {
      if (!tg->ev && hipEventCreateWithFlags(&tg->ev, hipEventDisableTiming) != hipSuccess)
        return qp_fail(QP_ERR_HIP, "qp_prune_tail: hipEventCreate failed");
      if (tg->pending && tg->stream != s && hipStreamWaitEvent(s, tg->ev, 0) != hipSuccess)
        return qp_fail(QP_ERR_HIP, "qp_prune_tail: hipStreamWaitEvent failed");
    }
    // launch 1: 16-bit norm key of every tail token (all KV heads of a token in one 16-lane group) + clears the sync words
    int rc = qp_launch_tail_keys(scored, head_stride, past_len, n, n_kv_heads, keys, largest, sync_words, (int)tail_slices(n), s);
    if (rc) return rc;
    // launch 2: radix select per 16-token slice, kept rows staged in registers, slice-ordered hand-shake, stores to [past, past+k)
    rc = qp_launch_prune_tail_inplace(keys, n, k, k_cache, v_cache, head_stride, past_len, n_kv_heads, kept_idx_out, sync_words, s);
    if (rc) return rc;
    if (!capturing) {
      if (hipEventRecord(tg->ev, s) != hipSuccess) return qp_fail(QP_ERR_HIP, "qp_prune_tail: hipEventRecord failed");
      tg->stream = s;
      tg->pending = true;
    }
    return QP_OK;
  }
  float* sumsq = (float*)ws;
  unsigned char* kt = ws + align256((size_t)n_kv_heads * n * 4);
  unsigned char* vt = kt + align256((size_t)n_kv_heads * k * head_dim * 2);
  int rc = qp_launch_key_sumsq(scored, head_stride, past_len, n, n_kv_heads, sumsq, s);
  if (rc) return rc;
  rc = qp_launch_select(sumsq, n_kv_heads, n, k, kept_idx_out, nullptr, nullptr, largest, s);
  if (rc) return rc;
  // gather the kept tail rows into the workspace, then copy them back contiguously (dst <= src row-wise, but
  // workgroups run in no defined order, so the in-place move goes through the scratch block)
  const unsigned short* kc = (const unsigned short*)k_cache;
  const unsigned short* vc = (const unsigned short*)v_cache;
  rc = qp_launch_gather_kv(kc + past_len * head_dim, vc + past_len * head_dim, head_stride, kept_idx_out, k, n_kv_heads, kt, vt,
                           k * head_dim, 0, s);
  if (rc) return rc;
  return qp_launch_copy_rows_kv(kt, vt, k * head_dim, k, n_kv_heads, k_cache, v_cache, head_stride, past_len, s);
}

int qp_sp_unpack(qp_ctx* ctx, const void* gathered, int world, int n_kv_heads, int64_t m2, int head_dim, int64_t n, void* k_stage,
                 void* v_stage, int64_t stage_head_stride, float* sumsq_out, void* stream) {
  QP_REQUIRE(ctx && gathered && k_stage && v_stage && sumsq_out, QP_ERR_INVALID, "qp_sp_unpack: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_sp_unpack: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(world >= 1 && n_kv_heads >= 1 && m2 >= 1 && n >= 0 && n <= 2 * (int64_t)world * m2, QP_ERR_INVALID,
             "qp_sp_unpack: n=%lld does not fit 2*world*m2=%lld rows", (long long)n, (long long)(2 * (int64_t)world * m2));
  QP_REQUIRE(stage_head_stride >= n * head_dim && stage_head_stride % 8 == 0, QP_ERR_INVALID, "qp_sp_unpack: stage_head_stride");
  QP_REQUIRE(aligned16(gathered) && aligned16(k_stage) && aligned16(v_stage), QP_ERR_INVALID, "qp_sp_unpack: alignment");
  QP_REQUIRE(((int64_t)n_kv_heads * 2 * m2 * 4) % 16 == 0, QP_ERR_INVALID, "qp_sp_unpack: per-rank block must be a multiple of 16 B");
  if (n == 0) return QP_OK;
  return qp_launch_sp_unpack(gathered, world, n_kv_heads, m2, n, k_stage, v_stage, stage_head_stride, sumsq_out, (hipStream_t)stream);
}

int qp_gather_rows(qp_ctx* ctx, const void* src, const int32_t* idx, int64_t k, int64_t row_bytes, void* dst, void* stream) {
  QP_REQUIRE(ctx && src && idx && dst, QP_ERR_INVALID, "qp_gather_rows: NULL argument");
  QP_REQUIRE(k >= 0 && row_bytes > 0 && row_bytes % 16 == 0, QP_ERR_INVALID, "qp_gather_rows: row_bytes=%lld must be a multiple of 16",
             (long long)row_bytes);
  QP_REQUIRE(aligned16(src) && aligned16(dst), QP_ERR_INVALID, "qp_gather_rows: alignment");
  if (k == 0) return QP_OK;
  return qp_launch_gather_rows(src, idx, k, row_bytes, dst, (hipStream_t)stream);
}

int qp_add_rmsnorm(qp_ctx* ctx, void* h, const void* delta, const void* w, void* out, int64_t n, int hidden, float eps,
                   void* stream) {
  QP_REQUIRE(ctx && h && w && out, QP_ERR_INVALID, "qp_add_rmsnorm: NULL argument");
  QP_REQUIRE(n >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 32768, QP_ERR_INVALID, "qp_add_rmsnorm: hidden=%d", hidden);
  QP_REQUIRE(aligned16(h) && aligned16(delta) && aligned16(w) && aligned16(out), QP_ERR_INVALID, "qp_add_rmsnorm: alignment");
  return qp_launch_add_rmsnorm(h, delta, w, out, n, hidden, eps, (hipStream_t)stream);
}

int qp_add_inplace(qp_ctx* ctx, void* h, const void* delta, int64_t n_elems, void* stream) {
  QP_REQUIRE(ctx && h && delta, QP_ERR_INVALID, "qp_add_inplace: NULL argument");
  QP_REQUIRE(n_elems >= 0 && n_elems % 8 == 0, QP_ERR_INVALID, "qp_add_inplace: n_elems must be a multiple of 8");
  QP_REQUIRE(aligned16(h) && aligned16(delta), QP_ERR_INVALID, "qp_add_inplace: alignment");
  if (n_elems == 0) return QP_OK;
  return qp_launch_add_inplace(h, delta, n_elems, (hipStream_t)stream);
}

int qp_swiglu(qp_ctx* ctx, const void* gate_up, int64_t n, int inter, void* out, void* stream) {
  QP_REQUIRE(ctx && gate_up && out, QP_ERR_INVALID, "qp_swiglu: NULL argument");
  QP_REQUIRE(n >= 0 && inter > 0 && inter % 8 == 0, QP_ERR_INVALID, "qp_swiglu: inter=%d must be a multiple of 8", inter);
  QP_REQUIRE(aligned16(gate_up) && aligned16(out), QP_ERR_INVALID, "qp_swiglu: alignment");
  if (n == 0) return QP_OK;
  return qp_launch_swiglu(gate_up, (const uint16_t*)gate_up + inter, 2ll * inter, n, inter, out, (hipStream_t)stream);
}

int qp_swiglu_split(qp_ctx* ctx, const void* gate, const void* up, int64_t n, int inter, void* out, void* stream) {
  QP_REQUIRE(ctx && gate && up && out, QP_ERR_INVALID, "qp_swiglu_split: NULL argument");
  QP_REQUIRE(n >= 0 && inter > 0 && inter % 8 == 0, QP_ERR_INVALID, "qp_swiglu_split: inter=%d must be a multiple of 8", inter);
  QP_REQUIRE(aligned16(gate) && aligned16(up) && aligned16(out), QP_ERR_INVALID, "qp_swiglu_split: alignment");
  if (n == 0) return QP_OK;
  return qp_launch_swiglu(gate, up, inter, n, inter, out, (hipStream_t)stream);
}

int qp_vit_rope(qp_ctx* ctx, void* qkv, const float* cos, const float* sin, int64_t n, int heads, int head_dim, void* stream) {
  QP_REQUIRE(ctx && qkv && cos && sin, QP_ERR_INVALID, "qp_vit_rope: NULL argument");
  QP_REQUIRE(n >= 0 && heads > 0 && head_dim > 0 && head_dim % 16 == 0, QP_ERR_INVALID, "qp_vit_rope: head_dim=%d must be a multiple of 16", head_dim);
  QP_REQUIRE(aligned16(qkv), QP_ERR_INVALID, "qp_vit_rope: alignment");
  if (n == 0) return QP_OK;
  return qp_launch_vit_rope(qkv, cos, sin, n, heads, head_dim, (hipStream_t)stream);
}

int qp_vit_attn(qp_ctx* ctx, const void* qkv, int64_t n_seq, int64_t seq_len, int heads, int head_dim, float scale, void* out,
                void* stream) {
  QP_REQUIRE(ctx && qkv && out, QP_ERR_INVALID, "qp_vit_attn: NULL argument");
  QP_REQUIRE(head_dim == 80, QP_ERR_UNSUPPORTED, "qp_vit_attn: head_dim=%d (only 80)", head_dim);
  QP_REQUIRE(n_seq >= 0 && seq_len >= 0 && heads > 0 && n_seq * heads <= 65535, QP_ERR_INVALID, "qp_vit_attn: bad sizes");
  QP_REQUIRE(seq_len * 3 * heads * head_dim * 2 < (1ll << 31), QP_ERR_UNSUPPORTED, "qp_vit_attn: sequence too long");
  QP_REQUIRE(aligned16(qkv) && aligned16(out), QP_ERR_INVALID, "qp_vit_attn: alignment");
  if (n_seq == 0 || seq_len == 0) return QP_OK;
  return qp_launch_vit_attn(ctx, qkv, n_seq, seq_len, heads, scale, out, nullptr, (hipStream_t)stream);
}

int qp_vit_attn_varlen(qp_ctx* ctx, const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t max_seq_len, int heads, int head_dim,
                       float scale, void* out, void* stream) {
  QP_REQUIRE(ctx && qkv && out && cu_seqlens, QP_ERR_INVALID, "qp_vit_attn_varlen: NULL argument");
  QP_REQUIRE(head_dim == 80, QP_ERR_UNSUPPORTED, "qp_vit_attn_varlen: head_dim=%d (only 80)", head_dim);
  QP_REQUIRE(n_seq >= 0 && max_seq_len >= 0 && heads > 0 && n_seq * heads < (1ll << 24), QP_ERR_INVALID, "qp_vit_attn_varlen: bad sizes");
  QP_REQUIRE(max_seq_len * 3 * heads * head_dim * 2 < (1ll << 31), QP_ERR_UNSUPPORTED, "qp_vit_attn_varlen: sequence too long");
  QP_REQUIRE(aligned16(qkv) && aligned16(out), QP_ERR_INVALID, "qp_vit_attn_varlen: alignment");
  if (n_seq == 0 || max_seq_len == 0) return QP_OK;
  return qp_launch_vit_attn(ctx, qkv, n_seq, max_seq_len, heads, scale, out, cu_seqlens, (hipStream_t)stream);
}

int qp_patchify(qp_ctx* ctx, const void* frames_u8, int n_frames, int height, int width, int patch, int temporal_patch, int merge,
                const void* lut_bf16, void* out, int row_elems, void* stream) {
  QP_REQUIRE(ctx && frames_u8 && lut_bf16 && out, QP_ERR_INVALID, "qp_patchify: NULL argument");
  QP_REQUIRE(patch > 0 && temporal_patch > 0 && merge > 0 && n_frames > 0 && n_frames % temporal_patch == 0 && height > 0 && width > 0 &&
                 height % (patch * merge) == 0 && width % (patch * merge) == 0,
             QP_ERR_INVALID, "qp_patchify: %d frames of %d x %d are not aligned to the patch grid (patch %d, temporal %d, merge %d)", n_frames, height,
             width, patch, temporal_patch, merge);
  QP_REQUIRE(row_elems >= 3 * temporal_patch * patch * patch && row_elems % 8 == 0 && aligned16(out), QP_ERR_INVALID,
             "qp_patchify: row_elems=%d (at least %d, a multiple of 8; out 16-byte aligned)", row_elems, 3 * temporal_patch * patch * patch);
  QP_REQUIRE((int64_t)n_frames * 3 * height * width < (1ll << 40), QP_ERR_UNSUPPORTED, "qp_patchify: frame block too large");
  return qp_launch_patchify(frames_u8, lut_bf16, out, n_frames, height, width, patch, temporal_patch, merge, row_elems, (hipStream_t)stream);
}

int qp_quick_gelu(qp_ctx* ctx, const void* x, void* out, int64_t n_elems, void* stream) {
  QP_REQUIRE(ctx && x && out, QP_ERR_INVALID, "qp_quick_gelu: NULL argument");
  QP_REQUIRE(n_elems >= 0 && n_elems % 8 == 0 && aligned16(x) && aligned16(out), QP_ERR_INVALID, "qp_quick_gelu: size/alignment");
  if (n_elems == 0) return QP_OK;
  return qp_launch_quick_gelu(x, out, n_elems, (hipStream_t)stream);
}

int qp_add_layernorm(qp_ctx* ctx, void* x, const void* delta, const void* w, const void* b, void* out, int64_t n, int hidden,
                     float eps, void* stream) {
  QP_REQUIRE(ctx && x && w && b && out, QP_ERR_INVALID, "qp_add_layernorm: NULL argument");
  QP_REQUIRE(n >= 0 && hidden > 0 && hidden % 8 == 0 && hidden <= 4096, QP_ERR_INVALID,
             "qp_add_layernorm: hidden=%d must be a multiple of 8, at most 4096", hidden);
  QP_REQUIRE(aligned16(x) && aligned16(w) && aligned16(b) && aligned16(out) && (!delta || aligned16(delta)), QP_ERR_INVALID,
             "qp_add_layernorm: alignment");
  return qp_launch_add_layernorm(x, delta, w, b, out, n, hidden, eps, (hipStream_t)stream);
}

int qp_linear_act(qp_ctx* ctx, const void* x, const void* w, const void* bias, int bias_f32, float alpha, void* out, int64_t m,
                  int64_t n, int64_t k, int act, void* workspace, size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && x && w && out, QP_ERR_INVALID, "qp_linear_act: NULL argument");
  QP_REQUIRE(m > 0 && n > 0 && k > 0 && k % 8 == 0 && n % 8 == 0, QP_ERR_INVALID, "qp_linear_act: m=%lld n=%lld k=%lld (n, k multiples of 8)",
             (long long)m, (long long)n, (long long)k);
  QP_REQUIRE(act >= 0 && act <= 1, QP_ERR_INVALID, "qp_linear_act: act=%d (0 none, 1 Swish z*sigmoid(z))", act);
  QP_REQUIRE(aligned16(x) && aligned16(w) && aligned16(out) && (!bias || aligned16(bias)) && (!workspace || aligned16(workspace)),
             QP_ERR_INVALID, "qp_linear_act: alignment");
  return qp_launch_linear_act(ctx, x, w, bias, bias_f32, alpha, out, m, n, k, act, workspace, workspace_bytes, (hipStream_t)stream);
}

int qp_linear_tune(qp_ctx* ctx, const void* x, const void* const* weights, int n_weights, const void* bias, int bias_f32, float alpha,
                   void* out, int64_t m, int64_t n, int64_t k, int act, void* workspace, size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && x && weights && out && n_weights > 0, QP_ERR_INVALID, "qp_linear_tune: NULL argument");
  QP_REQUIRE(m > 0 && n > 0 && k > 0 && k % 8 == 0 && n % 8 == 0, QP_ERR_INVALID, "qp_linear_tune: m=%lld n=%lld k=%lld", (long long)m,
             (long long)n, (long long)k);
  QP_REQUIRE(act >= 0 && act <= 1, QP_ERR_INVALID, "qp_linear_tune: act=%d", act);
  for (int i = 0; i < n_weights; ++i) QP_REQUIRE(weights[i] && aligned16(weights[i]), QP_ERR_INVALID, "qp_linear_tune: weight %d", i);
  return qp_launch_linear_tune(ctx, x, weights, n_weights, bias, bias_f32, alpha, out, m, n, k, act, workspace, workspace_bytes,
                               (hipStream_t)stream, nullptr);
}

int qp_linear_plan_choice(qp_ctx* ctx, int64_t m, int64_t n, int64_t k, int act, int bias_kind, int* choice, int* tuned) {
  QP_REQUIRE(ctx && choice, QP_ERR_INVALID, "qp_linear_plan_choice: NULL argument");
  *choice = -1;
  if (tuned) *tuned = 0;
  QP_REQUIRE(m > 0 && n > 0 && k > 0 && act >= 0 && act <= 1 && bias_kind >= 0 && bias_kind <= 2, QP_ERR_INVALID,
             "qp_linear_plan_choice: m=%lld n=%lld k=%lld act=%d bias_kind=%d", (long long)m, (long long)n, (long long)k, act, bias_kind);
  return qp_linear_plan_choice_impl(ctx, m, n, k, act, bias_kind, choice, tuned);
}

// ---- decode step (qp_decode.hip) ------------------------------------------------------------------------------------
int qp_gemv(qp_ctx* ctx, const void* w, const void* x, const void* norm_w, float eps, const void* bias, void* out,
            int64_t n_out, int64_t k, int mode, void* stream) {
  QP_REQUIRE(ctx && w && x && out, QP_ERR_INVALID, "qp_gemv: NULL argument");
  QP_REQUIRE(mode >= QP_GEMV_BIAS && mode <= QP_GEMV_RESIDUAL, QP_ERR_INVALID, "qp_gemv: unknown mode %d", mode);
  QP_REQUIRE(n_out > 0 && n_out < (1ll << 30) && k > 0 && k % 8 == 0 && k <= 32256, QP_ERR_INVALID,
             "qp_gemv: n_out=%lld, k=%lld (k must be a multiple of 8, at most 32256: x is staged in 63 KB of LDS)", (long long)n_out,
             (long long)k);
  QP_REQUIRE(mode == QP_GEMV_BIAS || bias == nullptr, QP_ERR_INVALID, "qp_gemv: bias only with QP_GEMV_BIAS");
  QP_REQUIRE(aligned16(w) && aligned16(x) && (!norm_w || aligned16(norm_w)), QP_ERR_INVALID, "qp_gemv: alignment");
  return qp_launch_gemv(ctx, w, x, norm_w, eps, bias, out, n_out, k, mode, (hipStream_t)stream);
}

int qp_decode_rope_append(qp_ctx* ctx, const void* qkv, const int64_t* state, const void* cos, const void* sin, float theta,
                          int n_q_heads, int n_kv_heads, int head_dim, void* q_out, void* k_cache, void* v_cache,
                          int64_t head_stride, void* stream) {
  QP_REQUIRE((cos == nullptr) == (sin == nullptr), QP_ERR_INVALID, "qp_decode_rope_append: cos and sin go together");
  QP_REQUIRE(!cos || (aligned16(cos) && aligned16(sin)), QP_ERR_INVALID, "qp_decode_rope_append: table alignment");
  QP_REQUIRE(ctx && qkv && state && q_out && k_cache && v_cache, QP_ERR_INVALID, "qp_decode_rope_append: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_decode_rope_append: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && head_stride >= 128 && head_stride % 8 == 0, QP_ERR_INVALID,
             "qp_decode_rope_append: heads/stride");
  QP_REQUIRE(aligned16(qkv) && aligned16(q_out) && aligned16(k_cache) && aligned16(v_cache), QP_ERR_INVALID,
             "qp_decode_rope_append: alignment");
  return qp_launch_decode_rope(qkv, state, cos, sin, theta, n_q_heads, n_kv_heads, q_out, k_cache, v_cache, head_stride,
                               (hipStream_t)stream);
}

size_t qp_decode_attn_workspace_bytes(const qp_ctx* ctx, int n_q_heads, int n_kv_heads) {
  if (!ctx || n_q_heads <= 0 || n_kv_heads <= 0) return 0;
  return qp_decode_attn_workspace_bytes_impl(ctx, n_q_heads, n_kv_heads);
}

int qp_decode_attn(qp_ctx* ctx, const void* q, const void* k_cache, const void* v_cache, int64_t head_stride,
                   const int64_t* state, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* out,
                   void* workspace, size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && q && k_cache && v_cache && state && out && workspace, QP_ERR_INVALID, "qp_decode_attn: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_decode_attn: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, QP_ERR_INVALID,
             "qp_decode_attn: n_q_heads=%d not a multiple of n_kv_heads=%d", n_q_heads, n_kv_heads);
  QP_REQUIRE(head_stride >= 128 && head_stride % 8 == 0, QP_ERR_INVALID, "qp_decode_attn: head stride");
  QP_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(workspace), QP_ERR_INVALID,
             "qp_decode_attn: alignment");
  const size_t need = qp_decode_attn_workspace_bytes_impl(ctx, n_q_heads, n_kv_heads);
  if (workspace_bytes < need) return qp_fail(QP_ERR_WORKSPACE, "qp_decode_attn: workspace %zu < %zu bytes", workspace_bytes, need);
  return qp_launch_decode_attn(ctx, q, k_cache, v_cache, head_stride, state, n_q_heads, n_kv_heads, scale, out, workspace,
                               (hipStream_t)stream);
}

int qp_decode_attn_fused(qp_ctx* ctx, const void* qkv, const void* cos, const void* sin, const int64_t* state, void* k_cache,
                         void* v_cache, int64_t head_stride, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* out,
                         void* workspace, size_t workspace_bytes, void* stream) {
  QP_REQUIRE(ctx && qkv && cos && sin && state && k_cache && v_cache && out && workspace, QP_ERR_INVALID,
             "qp_decode_attn_fused: NULL argument");
  QP_REQUIRE(head_dim == 128, QP_ERR_UNSUPPORTED, "qp_decode_attn_fused: head_dim=%d (only 128)", head_dim);
  QP_REQUIRE(n_q_heads > 0 && n_kv_heads > 0 && n_q_heads % n_kv_heads == 0, QP_ERR_INVALID,
             "qp_decode_attn_fused: n_q_heads=%d not a multiple of n_kv_heads=%d", n_q_heads, n_kv_heads);
  QP_REQUIRE(n_q_heads / n_kv_heads <= 8, QP_ERR_UNSUPPORTED, "qp_decode_attn_fused: at most 8 query heads per kv head (got %d)",
             n_q_heads / n_kv_heads);
  QP_REQUIRE(head_stride >= 128 && head_stride % 8 == 0, QP_ERR_INVALID, "qp_decode_attn_fused: head stride");
  QP_REQUIRE(aligned16(qkv) && aligned16(cos) && aligned16(sin) && aligned16(k_cache) && aligned16(v_cache) && aligned16(workspace),
             QP_ERR_INVALID, "qp_decode_attn_fused: alignment");
  const size_t need = qp_decode_attn_workspace_bytes_impl(ctx, n_q_heads, n_kv_heads);
  if (workspace_bytes < need) return qp_fail(QP_ERR_WORKSPACE, "qp_decode_attn_fused: workspace %zu < %zu bytes", workspace_bytes, need);
  return qp_launch_decode_attn_fused(ctx, qkv, cos, sin, k_cache, v_cache, head_stride, state, n_q_heads, n_kv_heads, scale, out,
                                     workspace, (hipStream_t)stream);
}

int qp_decode_advance(qp_ctx* ctx, int64_t* state, int64_t n_values, void* stream) {
  QP_REQUIRE(ctx && state, QP_ERR_INVALID, "qp_decode_advance: NULL argument");
  QP_REQUIRE(n_values > 0 && n_values <= (1 << 20), QP_ERR_INVALID, "qp_decode_advance: n_values=%lld", (long long)n_values);
  return qp_launch_decode_advance(state, (int)n_values, (hipStream_t)stream);
}

}  // extern "C"
