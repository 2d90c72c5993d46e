// qp_prefill_segment: one segment through all decoder layers in ONE call (include/quickprefill.h).  Host code only: it sequences the
// library's own launches — the same entry points, the same order and arguments as quickvideo_amd/engine.py::forward_segment issues
// them one by one — so a group costs one call from the caller's language instead of ~13 per layer (28 layers: 370 calls, 25-30 ms
// of interpreter time per group, each torch call among them handing the interpreter lock to whoever waits for it).
#include "qp_common.h"
#include <math.h>

namespace {

// out[m][n_out] = x[m][k] W^T (+ bias), optionally as two GEMMs over rows [0, split) and [split, m)
int linear(qp_ctx* ctx, const void* x, const void* w, const void* bias, void* out, int64_t m, int64_t n_out, int64_t k, int64_t split,
           void* ws, size_t ws_bytes, void* stream) {
  if (split <= 0 || split >= m) return qp_linear_act(ctx, x, w, bias, 0, 1.0f, out, m, n_out, k, 0, ws, ws_bytes, stream);
  int rc = qp_linear_act(ctx, x, w, bias, 0, 1.0f, out, split, n_out, k, 0, ws, ws_bytes, stream);
  if (rc) return rc;
  const char* x2 = (const char*)x + (size_t)split * k * 2;
  char* o2 = (char*)out + (size_t)split * n_out * 2;
  return qp_linear_act(ctx, x2, w, bias, 0, 1.0f, o2, m - split, n_out, k, 0, ws, ws_bytes, stream);
}

}  // namespace

extern "C" int qp_prefill_segment(qp_ctx* ctx, const qp_segment* g, const qp_layer* layers, int64_t* cache_len, const int64_t* k_keep,
                                  void* stream) {
  QP_REQUIRE(ctx && g && layers && cache_len && k_keep, QP_ERR_INVALID, "qp_prefill_segment: NULL argument");
  QP_REQUIRE(g->n_layers > 0 && g->n > 0 && g->hidden > 0 && g->intermediate > 0 && g->n_q_heads > 0 && g->n_kv_heads > 0, QP_ERR_INVALID,
             "qp_prefill_segment: bad sizes");
  QP_REQUIRE(g->head_dim == 128, QP_ERR_UNSUPPORTED, "qp_prefill_segment: head_dim=%d (only 128)", g->head_dim);
  QP_REQUIRE(g->h && g->x && g->qkv && g->q && g->att && g->o && g->gate_up && g->act && g->down && g->cos && g->sin && g->gemm_ws,
             QP_ERR_INVALID, "qp_prefill_segment: NULL buffer");
  const int64_t n = g->n, d = g->hidden, I = g->intermediate;
  const int hq = g->n_q_heads, hkv = g->n_kv_heads, D = g->head_dim;
  const int64_t hs = g->cache_capacity * D;                 // cache head stride in elements
  const int64_t qkv_cols = (int64_t)(hq + 2 * hkv) * D;
  bool prunes = false;
  for (int l = 0; l < g->n_layers; ++l) {
    QP_REQUIRE(k_keep[l] == -1 || (k_keep[l] > 0 && k_keep[l] <= n), QP_ERR_INVALID, "qp_prefill_segment: k_keep[%d]=%lld (n=%lld)", l,
               (long long)k_keep[l], (long long)n);
    QP_REQUIRE(cache_len[l] >= 0 && cache_len[l] + (k_keep[l] < 0 ? n : k_keep[l]) <= g->cache_capacity, QP_ERR_INVALID,
               "qp_prefill_segment: KV cache overflow in layer %d (%lld rows in use + %lld > capacity %lld)", l, (long long)cache_len[l],
               (long long)(k_keep[l] < 0 ? n : k_keep[l]), (long long)g->cache_capacity);
    prunes |= k_keep[l] >= 0;
  }
  if (prunes) {
    QP_REQUIRE(g->k_stage && g->v_stage && g->norm_keys && g->kept_idx && g->kept_idx_stride >= n, QP_ERR_INVALID,
               "qp_prefill_segment: pruning layers need k_stage / v_stage / norm_keys / kept_idx");
    QP_REQUIRE(g->prune_mode == QP_PRUNE_KEY_NORMS_SMALL || g->prune_mode == QP_PRUNE_KEY_NORMS, QP_ERR_UNSUPPORTED,
               "qp_prefill_segment: prune_mode=%d (key-row modes only; the other modes go through the per-operator entry points)", g->prune_mode);
    QP_REQUIRE(qp_rope_can_fuse_keys(hq, hkv) && n <= 8192, QP_ERR_UNSUPPORTED,
               "qp_prefill_segment: %d q / %d kv heads, n=%lld: norm keys cannot be fused into the RoPE kernel for this layout", hq, hkv, (long long)n);
  }
  hipStream_t s = (hipStream_t)stream;
  const void* delta = nullptr;
  int rc = QP_OK;
  for (int l = 0; l < g->n_layers; ++l) {
    const qp_layer& w = layers[l];
    const int64_t past = cache_len[l], kk = k_keep[l];
    // h += delta; x = RMSNorm(h)                                                                       (qwen25_lvu.py:167-169)
    if ((rc = qp_add_rmsnorm(ctx, g->h, delta, w.ln1, g->x, n, (int)d, g->rms_eps, stream))) return rc;
    // q/k/v projections + bias                                                                         (:42-44)
    if ((rc = linear(ctx, g->x, w.w_qkv, w.b_qkv, g->qkv, n, qkv_cols, d, g->split_qkv, g->gemm_ws, g->gemm_ws_bytes, stream))) return rc;
    const void *kn, *vn;
    int64_t new_stride;
    if (kk >= 0) {                                                // prune layer: new K/V to staging, 16-bit norm keys on the way
      rc = qp_rope_append_keys(ctx, g->qkv, g->cos, g->sin, n, hq, hkv, D, g->q, g->k_stage, g->v_stage, n * D, 0, nullptr, g->norm_keys,
                               g->prune_mode, stream);
      kn = g->k_stage; vn = g->v_stage; new_stride = n * D;
    } else {                                                      // append in place                     (:56-58)
      rc = qp_rope_append(ctx, g->qkv, g->cos, g->sin, n, hq, hkv, D, g->q, w.k_cache, w.v_cache, hs, past, nullptr, stream);
      kn = (const char*)w.k_cache + (size_t)past * D * 2; vn = (const char*)w.v_cache + (size_t)past * D * 2; new_stride = hs;
    }
    if (rc) return rc;
    // attention over (cache prefix, new rows), native GQA, bottom-right causal                         (:61-62, :102-112)
    const int64_t prefix = g->attend_prefix ? past : 0;
    const size_t need = qp_attn_workspace_bytes(ctx, n, prefix, hq, hkv);
    QP_REQUIRE(need <= g->attn_ws_bytes && (need == 0 || g->attn_ws), QP_ERR_WORKSPACE, "qp_prefill_segment: attention workspace %zu < %zu bytes",
               g->attn_ws_bytes, need);
    if (g->attn_events && hipEventRecord((hipEvent_t)g->attn_events[2 * l], s) != hipSuccess)
      return qp_fail(QP_ERR_HIP, "qp_prefill_segment: hipEventRecord failed");
    if ((rc = qp_prefill_attn(ctx, g->q, w.k_cache, w.v_cache, hs, prefix, kn, vn, new_stride, n, hq, hkv, D, g->attn_scale, g->att,
                              g->attn_ws, g->attn_ws_bytes, stream))) return rc;
    if (g->attn_events && hipEventRecord((hipEvent_t)g->attn_events[2 * l + 1], s) != hipSuccess)
      return qp_fail(QP_ERR_HIP, "qp_prefill_segment: hipEventRecord failed");
    // o_proj                                                                                            (:114-115)
    if ((rc = linear(ctx, g->att, w.w_o, nullptr, g->o, n, d, (int64_t)hq * D, g->split_o, g->gemm_ws, g->gemm_ws_bytes, stream))) return rc;
    // post_process_kv_cache: keep the kk smallest (largest) key norms, rows staging -> cache tail       (:183-192; utils.py:266-342)
    if (kk >= 0) {
      if (g->prune_events && hipEventRecord((hipEvent_t)g->prune_events[2 * l], s) != hipSuccess)
        return qp_fail(QP_ERR_HIP, "qp_prefill_segment: hipEventRecord failed");
      if ((rc = qp_prune_keys(ctx, g->norm_keys, n, kk, g->k_stage, g->v_stage, n * D, hkv, D, w.k_cache, w.v_cache, hs, past,
                              g->kept_idx + (int64_t)l * g->kept_idx_stride, stream))) return rc;
      if (g->prune_events && hipEventRecord((hipEvent_t)g->prune_events[2 * l + 1], s) != hipSuccess)
        return qp_fail(QP_ERR_HIP, "qp_prefill_segment: hipEventRecord failed");
      cache_len[l] = past + kk;
    } else {
      cache_len[l] = past + n;
    }
    // h += attn; x = RMSNorm(h)                                                                         (:182, :195-196)
    if ((rc = qp_add_rmsnorm(ctx, g->h, g->o, w.ln2, g->x, n, (int)d, g->rms_eps, stream))) return rc;
    // MLP: SiLU(gate) * up, down                                                                        (:197)
    if (g->gate_up_two_gemms) {
      char* gate = (char*)g->gate_up;
      char* up = gate + (size_t)n * I * 2;
      const char* w_up = (const char*)w.w_gate_up + (size_t)I * d * 2;
      if ((rc = linear(ctx, g->x, w.w_gate_up, nullptr, gate, n, I, d, g->split_gate_up, g->gemm_ws, g->gemm_ws_bytes, stream))) return rc;
      if ((rc = linear(ctx, g->x, w_up, nullptr, up, n, I, d, g->split_gate_up, g->gemm_ws, g->gemm_ws_bytes, stream))) return rc;
      if ((rc = qp_swiglu_split(ctx, gate, up, n, (int)I, g->act, stream))) return rc;
    } else {
      if ((rc = linear(ctx, g->x, w.w_gate_up, nullptr, g->gate_up, n, 2 * I, d, g->split_gate_up, g->gemm_ws, g->gemm_ws_bytes, stream))) return rc;
      if ((rc = qp_swiglu(ctx, g->gate_up, n, (int)I, g->act, stream))) return rc;
    }
    if ((rc = linear(ctx, g->act, w.w_down, nullptr, g->down, n, d, I, g->split_down, g->gemm_ws, g->gemm_ws_bytes, stream))) return rc;
    delta = g->down;
  }
  return qp_add_inplace(ctx, g->h, delta, n * d, stream);          // last residual                       (:198)
}

// The vision tower's blocks in one call (include/quickprefill.h: qp_vit_blocks) — same launches, same order as quickvideo_amd/vit.py.
extern "C" int qp_vit_blocks(qp_ctx* ctx, const qp_vit_block* blocks, int n_blocks, int64_t n_seq, int64_t seq_len, int dim, int heads,
                             int mlp_dim, void* x, void* y, void* qkv, void* att, void* pending, void* z, const float* cos, const float* sin,
                             float ln_eps, void* gemm_ws, size_t gemm_ws_bytes, void* stream) {
  QP_REQUIRE(ctx && blocks && x && y && qkv && att && pending && z && cos && sin && gemm_ws, QP_ERR_INVALID, "qp_vit_blocks: NULL argument");
  QP_REQUIRE(n_blocks > 0 && n_seq > 0 && seq_len > 0 && dim > 0 && heads > 0 && dim % heads == 0 && mlp_dim > 0, QP_ERR_INVALID,
             "qp_vit_blocks: bad sizes");
  const int64_t n = n_seq * seq_len;
  const int hd = dim / heads;
  const float scale = (float)pow((double)hd, -0.5);          // the value the per-operator caller passes: float(head_dim ** -0.5)
  const void* pend = nullptr;
  int rc = QP_OK;
  for (int b = 0; b < n_blocks; ++b) {
    const qp_vit_block& w = blocks[b];
    if ((rc = qp_add_layernorm(ctx, x, pend, w.ln1_w, w.ln1_b, y, n, dim, ln_eps, stream))) return rc;
    if ((rc = qp_linear_act(ctx, y, w.qkv_w, w.qkv_b, 0, 1.0f, qkv, n, 3 * (int64_t)dim, dim, 0, gemm_ws, gemm_ws_bytes, stream))) return rc;
    if ((rc = qp_vit_rope(ctx, qkv, cos, sin, n, heads, hd, stream))) return rc;
    if ((rc = qp_vit_attn(ctx, qkv, n_seq, seq_len, heads, hd, scale, att, stream))) return rc;
    if ((rc = qp_linear_act(ctx, att, w.proj_w, w.proj_b, 0, 1.0f, pending, n, dim, dim, 0, gemm_ws, gemm_ws_bytes, stream))) return rc;
    if ((rc = qp_add_layernorm(ctx, x, pending, w.ln2_w, w.ln2_b, y, n, dim, ln_eps, stream))) return rc;
    if ((rc = qp_linear_act(ctx, y, w.fc1_w, w.fc1_bias_scaled, 1, 1.702f, z, n, mlp_dim, dim, 1, gemm_ws, gemm_ws_bytes, stream))) return rc;
    if ((rc = qp_linear_act(ctx, z, w.fc2_w, w.fc2_b, 0, 1.0f / 1.702f, pending, n, dim, mlp_dim, 0, gemm_ws, gemm_ws_bytes, stream))) return rc;
    pend = pending;
  }
  return QP_OK;
}
