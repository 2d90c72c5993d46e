/*
 * quickprefill.h — C ABI of libquickprefill.so, the MI355X (gfx950) QuickPrefill hot path.
 *
 * One entry point per operator seam of the reference's group-chunked prefill with key-L2-norm
 * KV-cache pruning (TIGER-AI-Lab/QuickVideo; citations are into /root/reference):
 *
 *   seam 1  prune     lvu/utils.py:197-376  post_process_kv_cache  (+ :133-136, :190-194 scoring)
 *   seam 2  append    lvu/models/qwen25_lvu.py:51-58 (M-RoPE + cache.update) -> lvu/lvu_cache.py:90-98
 *   seam 3  attention lvu/models/qwen25_lvu.py:61-62,102-112 (repeat_kv + flash_attn causal)
 *   glue    RMSNorm / SwiGLU of the patched decoder layer, qwen25_lvu.py:169,196-198 [transformers]
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host".
 *   - bf16 tensors are passed as void* (16-bit patterns); the caller owns every buffer, including the
 *     workspace sized with qp_select_workspace_bytes(); the library allocates nothing on the device
 *     and never synchronises: all work is enqueued on the hipStream_t given (passed as void*).
 *   - every function returns QP_OK (0) or a negative qp_status; qp_last_error() gives a thread-local
 *     message.  Invalid arguments are rejected before anything is launched.
 *   - KV arena layout per layer: K and V are [n_kv_heads][capacity][head_dim] bf16; "head stride"
 *     arguments are in ELEMENTS (capacity*head_dim for the arena).  head_dim must be 128.
 *   - "new" K/V of the group being prefilled may live either in the arena tail
 *     (k_new = k_cache + prefix_len*head_dim, same head stride) or in a separate staging block
 *     [n_kv_heads][n][head_dim]; attention takes (prefix, new) as two segments so the pruning step is a
 *     pure gather staging -> arena with no in-place hazard.
 */
#ifndef QUICKPREFILL_H_
#define QUICKPREFILL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qp_ctx qp_ctx;

typedef enum qp_status {
  QP_OK = 0,
  QP_ERR_INVALID = -1,     /* bad argument (maps to ValueError / AssertionError in the Python mirror) */
  QP_ERR_UNSUPPORTED = -2, /* shape outside what the kernels implement                               */
  QP_ERR_HIP = -3,         /* a HIP runtime call failed                                              */
  QP_ERR_WORKSPACE = -4,   /* workspace too small                                                    */
  QP_ERR_TIMEOUT = -5      /* a bounded wait ran out (qp_frame_ring_acquire_for): not a failure, call again */
} qp_status;

/* ---- context ------------------------------------------------------------------------------ */
/* device: HIP ordinal.  One context per device/process (one process per GPU under tensor parallel). */
int qp_create(qp_ctx** out, int device);
/* Developer A/B switches (tools/bench_attn.py, the kernel-form parity tests): "attn_variant" (0 production; 1 v1, 2 no kv split, 3 no
 * XCD map, 4 s4, 7 / 8 = 4- / 8-wave s6), "attn_force_split", "attn_flat" (-1 by cost, 0 never, 1 whenever eligible), "s6_prio", "s6_early_out", "decode_attn_valu", "attn_debug".
 * Process-wide.
 * The launch paths never read the environment: the table is filled once at first use from QP_ATTN_VARIANT, QP_ATTN_FORCE_SPLIT,
 * QP_ATTN_FLAT, QP_S6_PRIO, QP_S6_EARLY_OUT, QP_DECODE_ATTN, QP_ATTN_DEBUG and changes only through this call.  Not part of the reference seams. */
int qp_dev_switch(const char* name, int value);
void qp_destroy(qp_ctx* ctx);
const char* qp_last_error(void);
const char* qp_version(void);          /* "quickprefill-mi355x 0.7 (gfx950)": 0.3 prune_mode became a per-call argument; 0.4 qp_prefill_segment; 0.5 qp_linear_plan_choice; 0.6 qp_frame_ring_*; 0.7 qp_linear_plan_choice returns a status, qp_frame_ring_acquire_for, qp_patchify */
int qp_device_cus(const qp_ctx* ctx);

/* Host helper of the overlap producer (no device work): memcpy `bytes` from src to dst (e.g. decoded uint8 frames into a pinned
 * ring slot), split over up to `threads` std::threads.  Callable with the GIL released (ctypes / cgo drop it around foreign
 * calls) and independent of torch's process-wide intra-op thread setting; the reference's producer thread fills its queue with
 * `.float()` + the HF processor under the GIL (qwen25_lvu_interleaved.py:303-340). */
int qp_host_memcpy(void* dst, const void* src, size_t bytes, int threads);

/* ---- seam 2: M-RoPE + KV append  (qwen25_lvu.py:46-58) -------------------------------------- */
/* cos/sin tables for one group: pos int64 [3][n] (temporal, height, width streams of get_rope_index,
 * qwen25_lvu.py:613-619, 689); sections = mrope_section (e.g. {16,24,24}, sum = head_dim/2);
 * out cos/sin bf16 [n][head_dim/2] (second half of the head repeats the first). */
int qp_mrope_table(qp_ctx* ctx, const int64_t* pos, int64_t n, const int32_t sections[3], float theta,
                   int head_dim, void* cos_out, void* sin_out, void* stream);

/* qkv bf16 [n][(n_q+2*n_kv)*head_dim] (q heads, then k heads, then v heads — the fused q/k/v
 * projection incl. bias).  Rotates q and k (rotate-half, bf16 rounding after each op like the
 * reference's bf16 tensors), writes q to q_out [n][n_q][head_dim] (must not overlap qkv), k/v to
 * k_dst/v_dst[h*dst_head_stride + (dst_row0+t)*head_dim ...], and — when head_sumsq != NULL — the
 * canonical per-head fp32 sum of squares of the STORED (bf16) key row to head_sumsq[h*n + t]. */
int qp_rope_append(qp_ctx* ctx, const void* qkv, const void* cos, const void* sin, int64_t n, int n_q_heads,
                   int n_kv_heads, int head_dim, void* q_out, void* k_dst, void* v_dst, int64_t dst_head_stride,
                   int64_t dst_row0, float* head_sumsq, void* stream);

/* Which rows the prune step (seam 1) keeps — the reference's norm-based `top_k_predict_type`s (utils.py:117-136).  A per-call
 * argument of every entry point whose result depends on it (no hidden context state): bit 0 = keep the k LARGEST norms (argsort
 * descending) instead of the k smallest, bit 1 = score the VALUE rows instead of the key rows.  Ties always resolve to the lowest
 * index.  Only qp_prune_tail reads bit 1 (it fetches the rows itself); the other entry points take sums the caller computed over
 * the rows of its choice (qp_key_sumsq) and use bit 0 alone. */
typedef enum qp_prune_mode {
  QP_PRUNE_KEY_NORMS_SMALL = 0,    /* "key_norms_small" (the reference's default)  */
  QP_PRUNE_KEY_NORMS = 1,          /* "key_norms"                                  */
  QP_PRUNE_VECTOR_NORMS_SMALL = 2, /* "vector_norms_small"                         */
  QP_PRUNE_VECTOR_NORMS = 3        /* "vector_norms"                               */
} qp_prune_mode;

/* qp_rope_append that also prepares the layer's prune (seam 1) while it holds the key rows: the 16-bit norm key of every token
 * (bf16 pattern of the cross-head key norm bf16(sqrt(((s0+s1)+s2)+...)), complemented when prune_mode says "k largest";
 * prune_mode must be one of the two KEY-row modes) goes to norm_keys[t] (uint16 [n]) for qp_prune_keys.  Needs all KV heads of the layer in this call (no tensor-parallel head
 * sharding) and n_kv_heads in {1, 2, 4} with n_q_heads a multiple of it; returns QP_ERR_UNSUPPORTED otherwise — callers then
 * use qp_rope_append + qp_norm_keys.  head_sumsq may be NULL. */
int qp_rope_append_keys(qp_ctx* ctx, const void* qkv, const void* cos, const void* sin, int64_t n, int n_q_heads,
                        int n_kv_heads, int head_dim, void* q_out, void* k_dst, void* v_dst, int64_t dst_head_stride,
                        int64_t dst_row0, float* head_sumsq, uint16_t* norm_keys, int prune_mode, void* stream);

/* ---- seam 3: prefill attention over (pruned prefix, new group)  (qwen25_lvu.py:61-62,102-112) - */
/* q bf16 [n][n_q][128]; prefix K/V rows [0,prefix_len) with head stride prefix_head_stride; new K/V rows
 * [0,n) with head stride new_head_stride.  Query i attends every prefix key and new keys j <= i
 * (flash-attn's bottom-right aligned causal mask); GQA native (q head h uses kv head h/(n_q/n_kv)).
 * out bf16 [n][n_q][128].  softmax(scale * q.k) in fp32, P rounded to bf16 before P.V. */
int qp_prefill_attn(qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix,
                    int64_t prefix_head_stride, int64_t prefix_len, const void* k_new, const void* v_new,
                    int64_t new_head_stride, int64_t n, int n_q_heads, int n_kv_heads, int head_dim, float scale,
                    void* out, void* workspace, size_t workspace_bytes, void* stream);
/* Same, for a sub-range of the group's queries (group-token parallel ranks: every rank holds all n new K/V rows but only
 * the queries [q_row0, q_row0+nq)): q and out are [nq][n_q][128]; query i of the sub-range attends new keys j <= q_row0+i. */
int qp_prefill_attn_rows(qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix,
                         int64_t prefix_head_stride, int64_t prefix_len, const void* k_new, const void* v_new,
                         int64_t new_head_stride, int64_t n, int64_t q_row0, int64_t nq, int n_q_heads, int n_kv_heads,
                         int head_dim, float scale, void* out, void* workspace, size_t workspace_bytes, void* stream);
/* Scratch for the kv-split partial results (work items of a ragged last round are cut along KV and merged by a combine
 * kernel so every CU stays busy; small grids — few heads x few query blocks — are split the same way).  `n` = number of
 * query rows of the launch (nq for qp_prefill_attn_rows) and `prefix_len` = keys every query sees before its own block
 * (prefix_len + q_row0 for a sub-range): the same pair the launch plans with.  workspace may be NULL: the launch then runs
 * unsplit. */
size_t qp_attn_workspace_bytes(const qp_ctx* ctx, int64_t n, int64_t prefix_len, int n_q_heads, int n_kv_heads);

/* ---- seam 1: key-norm scoring, k-smallest select, compaction  (utils.py:133-136, 266-342) ---- */
/* Canonical per-head sum of squares of key rows k[h*head_stride + (row0+t)*head_dim ...], t<n ->
 * head_sumsq[h*n+t] fp32.  (Unfused variant of what qp_rope_append emits.) */
int qp_key_sumsq(qp_ctx* ctx, const void* k, int64_t head_stride, int64_t row0, int64_t n, int n_kv_heads,
                 int head_dim, float* head_sumsq, void* stream);

size_t qp_select_workspace_bytes(int64_t n);

/* head_sumsq fp32 [n_heads_total][n] (all KV heads of the layer, ascending head order; under tensor
 * parallelism the all-gathered per-rank partials).  norm[t] = bf16(sqrt(((s0+s1)+s2)+...)).
 * kept_idx_out int32 [k]: the k smallest norms (k largest when bit 0 of prune_mode is set), ties -> lowest index, listed in
 * ascending index order (utils.py:136,191-194,284).  norm_bits_out (uint16 [n], may be NULL) receives the bf16 norms.
 * Requires 0 < k <= n.  n <= 65536: norms stay in LDS, workspace may be NULL; larger n (single-group baseline mode on
 * long videos): pass a workspace of qp_select_workspace_bytes(n). */
int qp_select_k_smallest(qp_ctx* ctx, const float* head_sumsq, int n_heads_total, int64_t n, int64_t k,
                         int32_t* kept_idx_out, uint16_t* norm_bits_out, int prune_mode, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The same select (utils.py:55-57 / 133-136: argsort of the scores, first k; :190-194 the mask) on ready-made 16-bit sort keys (uint16 [n]: what qp_rope_append_keys / qp_norm_keys / qp_query_scores emit — bf16
 * norm patterns, complemented where "largest" is wanted): the k smallest keys, ties -> lowest index, ascending index list.  Any n
 * (keys stay in LDS up to 65536, are streamed from where they are beyond); with qp_gather_kv this is the prune step of groups
 * beyond qp_prune_keys' 8192 tokens when the keys already exist (query-score mode, fused RoPE keys). */
int qp_select_keys(qp_ctx* ctx, const uint16_t* norm_keys, int64_t n, int64_t k, int32_t* kept_idx_out, void* stream);

/* dst[h*dst_head_stride + (dst_row0+j)*head_dim ...] = src[h*src_head_stride + idx[j]*head_dim ...]
 * for j<k, for K and V (utils.py:287-288, 333-336).  src and dst must not overlap. */
int qp_gather_kv(qp_ctx* ctx, const void* k_src, const void* v_src, int64_t src_head_stride, const int32_t* idx,
                 int64_t k, int n_kv_heads, int head_dim, void* k_dst, void* v_dst, int64_t dst_head_stride,
                 int64_t dst_row0, void* stream);

/* qp_select_k_smallest + qp_gather_kv behind one call (two launches), for groups beyond qp_prune_keys' 8192 tokens (the
 * single-group baseline mode of long videos); kept_idx_out / norm_bits_out as above. */
int qp_prune_staged(qp_ctx* ctx, const float* head_sumsq, int n_heads_total, int64_t n, int64_t k, const void* k_src,
                    const void* v_src, int64_t src_head_stride, int n_kv_heads, int head_dim, void* k_dst, void* v_dst,
                    int64_t dst_head_stride, int64_t dst_row0, int32_t* kept_idx_out, uint16_t* norm_bits_out, int prune_mode,
                    void* stream);

/* The engine's prune step since round 2 (one launch):
 *   qp_norm_keys   head_sumsq fp32 [n_heads_total][n] -> norm_keys uint16 [n].  Only needed when qp_rope_append_keys could not
 *                  produce them (tensor-/group-token-parallel gathered sums, value-row norms, 8 KV heads in one process).
 *   qp_prune_keys  one workgroup per 16 tokens: all n keys in registers, threshold key by a two-pass radix select in LDS, kept
 *                  tokens in front of the slice counted from the same registers, K/V rows of the slice's kept tokens moved
 *                  src -> dst rows [dst_row0, dst_row0+k); kept_idx_out int32 [k] ascending; same tie rule as
 *                  qp_select_k_smallest (keys below the threshold, then ties by lowest index).  Requires 0 < k <= n <= 8192,
 *                  n_kv_heads <= 8 (larger n: qp_select_k_smallest + qp_gather_kv). */
int qp_norm_keys(qp_ctx* ctx, const float* head_sumsq, int n_heads_total, int64_t n, uint16_t* norm_keys, int prune_mode,
                 void* stream);
int qp_prune_keys(qp_ctx* ctx, const uint16_t* norm_keys, int64_t n, int64_t k, const void* k_src, const void* v_src,
                  int64_t src_head_stride, int n_kv_heads, int head_dim, void* k_dst, void* v_dst, int64_t dst_head_stride,
                  int64_t dst_row0, int32_t* kept_idx_out, void* stream);

/* Query-attention-score pruning (top_k_predict_type "query_attention_weights[_by_value_norm]"; LVUCache.update in query-based
 * mode, lvu_cache.py:97-117, and utils.py:55-62): q_prompt bf16 [m][n_q][128] = the RoPE'd queries of the m prompt tokens the
 * reference appends to every video group (qwen25_lvu.py:684-686), k_group = the group's own RoPE'd keys [n_kv][n][128] (head
 * stride k_head_stride).  score[t] = bf16(mean_h bf16(sum_q bf16(softmax_fp32(bf16(bf16(q.k)/sqrt(128))))))  -> scores_out
 * (uint16 [n], may be NULL); norm_keys_out[t] = ~pattern(score) — or ~pattern(bf16(score * ||v_t||)) when value_sumsq (fp32
 * [n_kv][n], from qp_key_sumsq over the value rows) is given — ready for qp_prune_keys: k largest scores, ties -> lowest index. */
size_t qp_query_scores_workspace_bytes(int64_t n, int64_t m, int n_q_heads);
/* The same scoring in two steps, for hosts that shard the heads (tensor parallelism): (1) per-head sums s1[h][t] = bf16(sum over the prompt
 * queries of the bf16 softmax probabilities) for the LOCAL heads (same workspace as qp_query_scores); (2) after the ranks' blocks were
 * gathered in ascending head order: mean over all heads, optional value-norm weighting (value_sumsq fp32 [n_kv_heads_total][n]),
 * complemented sort keys.  qp_query_scores == (1) then (2) on one device, bit for bit (it IS the composition). */
int qp_query_head_sums(qp_ctx* ctx, const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int n_q_heads,
                       int n_kv_heads, int head_dim, uint16_t* head_sums_out, void* workspace, size_t workspace_bytes, void* stream);
int qp_query_scores_from_head_sums(qp_ctx* ctx, const uint16_t* head_sums, int n_heads_total, int64_t n, const float* value_sumsq,
                                   int n_kv_heads_total, uint16_t* norm_keys_out, uint16_t* scores_out, void* stream);
int qp_query_scores(qp_ctx* ctx, const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m,
                    int n_q_heads, int n_kv_heads, int head_dim, const float* value_sumsq, uint16_t* norm_keys_out,
                    uint16_t* scores_out, void* workspace, size_t workspace_bytes, void* stream);

/* In-place drop-in for post_process_kv_cache's KV part on the arena (utils.py:266-342):
 * rows [past_len, past_len+n) are the group's new tokens; on return rows [past_len, past_len+k) hold the
 * kept ones in original order and kept_idx_out[k] lists them.  workspace >= qp_prune_workspace_bytes().
 * Two launches for n <= 8192 and n_kv_heads <= 8 (round 3): (1) the 16-bit norm key of every tail token; (2) one workgroup per
 * 16-token slice finds the threshold by a radix select in LDS, STAGES the K/V rows of its kept tokens in registers (<= 256 B per
 * lane, <= 64 KB per workgroup), signals "rows loaded", waits for the (at most two, always LOWER) slices whose source rows its
 * destination rows overlap, and stores to [past_len, past_len+k) — the compaction never bounces through HBM scratch.  The wait is
 * deadlock-free because it only points downwards, the whole grid (<= 512 workgroups) fits at once on the CUs the stream may use
 * (occupancy API x the stream's CU mask; otherwise the staged form runs) and the library keeps at most ONE in-place grid in flight PER DEVICE
 * (whichever context or stream launches it): a call on a different stream than the previous one is ordered behind it on the device
 * (hipStreamWaitEvent; no host wait).  Callers may therefore use any number of streams and contexts.  (Inside a stream capture the library records no events: the captured stream orders its
 * own nodes.)  The in-place workspace only holds the keys and the flags (2.25 bytes per token).  Larger groups use the round-1 form
 * (sums -> select -> gather into the workspace -> copy back), which needs room for the kept rows.
 *   qp_prune_workspace_bytes(...)            an upper bound, sufficient for either form on any device / stream (no context needed)
 *   qp_prune_tail_workspace_bytes(ctx, ...)  the exact size for this context and stream (NULL = an unmasked stream): the small
 *                                            in-place figure only when qp_prune_tail will take that form there */
size_t qp_prune_workspace_bytes(int64_t n, int64_t k, int n_kv_heads, int head_dim);
size_t qp_prune_tail_workspace_bytes(const qp_ctx* ctx, int64_t n, int64_t k, int n_kv_heads, int head_dim, void* stream);
int qp_prune_tail(qp_ctx* ctx, void* k_cache, void* v_cache, int64_t head_stride, int64_t past_len, int64_t n,
                  int64_t k, int n_kv_heads, int head_dim, int32_t* kept_idx_out, int prune_mode, void* workspace,
                  size_t workspace_bytes, void* stream);

/* Row gather for the hidden-state pruning hand-off (prune_for_next_layer; utils.py:292-331):
 * dst[j][:] = src[idx[j]][:], rows of row_bytes (multiple of 16). */
int qp_gather_rows(qp_ctx* ctx, const void* src, const int32_t* idx, int64_t k, int64_t row_bytes, void* dst,
                   void* stream);

/* Group-token parallel exchange, receive side (no reference counterpart: the reference is single-GPU; this is the multi-GPU
 * form of cache.update, qwen25_lvu.py:56-58).  `gathered` = the all-gather of every rank's send block, [world][chunk]:
 * chunk = K bf16 [n_kv][2*m2][head_dim] | V same | key sums fp32 [n_kv][2*m2]; rank r's rows are the two zigzag chunks
 * r and 2*world-1-r of the group's n tokens (m2 = ceil(n / (2*world)) rows each).  Writes K/V rows in TOKEN order to
 * k_stage/v_stage[h*stage_head_stride + t*head_dim ...] and the sums to sumsq_out[h*n + t], t < n. */
int qp_sp_unpack(qp_ctx* ctx, const void* gathered, int world, int n_kv_heads, int64_t m2, int head_dim, int64_t n,
                 void* k_stage, void* v_stage, int64_t stage_head_stride, float* sumsq_out, void* stream);

/* ---- glue of the patched decoder layer (qwen25_lvu.py:166-198) [transformers Qwen2RMSNorm/MLP] -- */
/* if delta != NULL: h = bf16(h + delta) (written back);  out = w * bf16(h * rsqrt(mean(h^2) + eps)). */
int qp_add_rmsnorm(qp_ctx* ctx, void* h, const void* delta, const void* w, void* out, int64_t n, int hidden,
                   float eps, void* stream);
/* h = bf16(h + delta) only (last residual of the layer). */
int qp_add_inplace(qp_ctx* ctx, void* h, const void* delta, int64_t n_elems, void* stream);
/* gate_up bf16 [n][2*inter] (gate columns then up columns) -> out[n][inter] = bf16(bf16(silu(g)) * u). */
int qp_swiglu(qp_ctx* ctx, const void* gate_up, int64_t n, int inter, void* out, void* stream);
/* Same with the two projections in separate [n][inter] buffers (hipBLASLt runs two N = inter GEMMs faster than one N = 2*inter
 * GEMM for some row counts on this hardware; the engine times both once per segment size). */
int qp_swiglu_split(qp_ctx* ctx, const void* gate, const void* up, int64_t n, int inter, void* out, void* stream);

/* ---- decode step over the pruned cache (qwen25_lvu.py:744-761: HF generate with the LVU cache) -------------------
 * Every per-token scalar lives in a DEVICE state block  state int64[2] = { kv_len, rope_pos }  (rows already in every
 * layer's cache; position id of the token being decoded, the same on all three M-RoPE streams), so a whole decode step has
 * no host arguments that change between tokens: capture it once in a hipGraph and replay it per token.              */
/* out[n_out] = epilogue(W[n_out][k] . x[k]) for ONE token (batch 1): the weight stream of the step, HBM-bound.
 * W bf16 row-major [n_out][k] (a torch Linear weight), k % 8 == 0, k <= 32256 (x is staged in LDS).  If norm_w != NULL, x is the residual
 * stream h and the kernel applies RMSNorm first (x = norm_w * bf16(h * rsqrt(mean(h^2) + eps)), same arithmetic as
 * qp_add_rmsnorm).  mode QP_GEMV_BIAS: out = bf16(dot + bias) (bias may be NULL);  QP_GEMV_SWIGLU: W is [2*n_out][k]
 * (gate rows then up rows), out = bf16(bf16(silu(bf16(g))) * bf16(u)) = qp_swiglu of the two projections;
 * QP_GEMV_RESIDUAL: out = bf16(out + bf16(dot)) in place (the residual add of the layer). */
enum { QP_GEMV_BIAS = 0, QP_GEMV_SWIGLU = 1, QP_GEMV_RESIDUAL = 2 };
int qp_gemv(qp_ctx* ctx, const void* w, const void* x, const void* norm_w, float eps, const void* bias, void* out,
            int64_t n_out, int64_t k, int mode, void* stream);
/* M-RoPE of the token's q and k and append of its K/V at row state[0] of the cache.  cos/sin: the bf16 [64] table of the
 * token's position from qp_mrope_table (computed once per token, shared by all layers; its `pos` argument is a device
 * pointer, so it replays too), or both NULL: computed in the kernel with the same arithmetic at position state[1].
 * qkv bf16 [(n_q+2*n_kv)*128]; q_out bf16 [n_q][128]. */
int qp_decode_rope_append(qp_ctx* ctx, const void* qkv, const int64_t* state, const void* cos, const void* sin, float theta,
                          int n_q_heads, int n_kv_heads, int head_dim, void* q_out, void* k_cache, void* v_cache,
                          int64_t head_stride, void* stream);
/* Single-query attention over cache rows [0, state[0]] (the token's own row included), GQA native (at most 8 query heads
 * per kv head: they are the query columns of one MFMA tile); fixed grid independent of the cache length.  out bf16 [n_q][128]. */
size_t qp_decode_attn_workspace_bytes(const qp_ctx* ctx, int n_q_heads, int n_kv_heads);
int qp_decode_attn(qp_ctx* ctx, const void* q, const void* k_cache, const void* v_cache, int64_t head_stride,
                   const int64_t* state, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* qp_decode_rope_append + qp_decode_attn in ONE launch (every kernel boundary costs ~4 us of a ~3.4 ms step): qkv is the RAW fused
 * projection, cos/sin the token's qp_mrope_table; the new K/V row is written to the cache by the workgroup that reads it. */
int qp_decode_attn_fused(qp_ctx* ctx, const void* qkv, const void* cos, const void* sin, const int64_t* state, void* k_cache,
                         void* v_cache, int64_t head_stride, int n_q_heads, int n_kv_heads, int head_dim, float scale, void* out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* state[i] += 1 for i < n_values (end of the step; the state blocks of all layers are one array: layers may hold
 * different numbers of rows under the decaying keep ratios, utils.py:231-251). */
int qp_decode_advance(qp_ctx* ctx, int64_t* state, int64_t n_values, void* stream);

/* ---- vision front end (transformers Qwen2VisionTransformerPretrainedModel [3P]) ---------------
 * The ViT tower itself runs on PyTorch-ROCm (quickvideo_amd/vit.py); these three entry points replace its non-GEMM ops.
 * qkv bf16 [n][3][heads][head_dim] = output of the fused qkv projection of one block (n = n_seq * S tokens).        */
/* 2-D rotary of q and k in place (apply_rotary_pos_emb_vision: fp32 math, one rounding).  cos/sin fp32 [n][head_dim/2]. */
int qp_vit_rope(qp_ctx* ctx, void* qkv, const float* cos, const float* sin, int64_t n, int heads, int head_dim, void* stream);
/* Full (non-causal) attention inside each of the n_seq sequences of S tokens (one per temporal patch: cu_seqlens of the
 * reference), head_dim 80, MFMA kernel shared with qp_prefill_attn.  out bf16 [n][heads][80]. */
int qp_vit_attn(qp_ctx* ctx, const void* qkv, int64_t n_seq, int64_t seq_len, int heads, int head_dim, float scale, void* out,
                void* stream);
/* Same for a ragged batch — the window attention of the Qwen2.5-VL tower (28 of its 32 blocks; transformers
 * get_vision_window_index [3P]): sequence i = rows [cu_seqlens[i], cu_seqlens[i+1]) of the packed, window-major qkv tensor
 * (cu_seqlens int32 [n_seq+1] on the device, lengths <= max_seq_len), full attention inside every window. */
int qp_vit_attn_varlen(qp_ctx* ctx, const void* qkv, const int32_t* cu_seqlens, int64_t n_seq, int64_t max_seq_len, int heads,
                       int head_dim, float scale, void* out, void* stream);
/* Residual add of a vision block fused with the LayerNorm after it: if delta != NULL, x = bf16(x + delta) (written back);
 * out = bf16((x - mean) * rstd * w + b) with fp32 statistics over the row (torch layer_norm).  x, delta, out bf16 [n][hidden]. */
int qp_add_layernorm(qp_ctx* ctx, void* x, const void* delta, const void* w, const void* b, void* out, int64_t n, int hidden,
                     float eps, void* stream);
/* out[m][n] = act(alpha * x[m][k] W[n][k]^T + bias[n]) as ONE hipBLASLt GEMM with the activation in its epilogue (act: 0 none,
 * 1 Swish z*sigmoid(z) = SiLU): fp32 accumulate + bias + activation, one rounding to bf16.  bias: bf16, or fp32 when bias_f32,
 * or NULL.  quick-GELU (y*sigmoid(1.702 y), the vision MLP's fc1): alpha = 1.702, bias pre-scaled by 1.702, act = 1 gives
 * 1.702*quick_gelu(y); the 1/1.702 goes onto the alpha of the next GEMM.  workspace: caller-owned scratch for hipBLASLt (128 MB
 * covers every shape tried; the call fails with QP_ERR_WORKSPACE if the chosen algorithm wants more).  ONE BUFFER PER STREAM: stream-K /
 * split-K algorithms keep partial tiles and flags there, so two GEMMs that may run concurrently (ViT stream and prefill stream) must
 * not share it — a shared buffer corrupts both and can leave a workgroup waiting on a flag forever. */
int qp_linear_act(qp_ctx* ctx, const void* x, const void* w, const void* bias, int bias_f32, float alpha, void* out, int64_t m,
                  int64_t n, int64_t k, int act, void* workspace, size_t workspace_bytes, void* stream);
/* Picks the hipBLASLt algorithm later qp_linear_act calls of this (m, n, k, act, bias kind) use: every heuristic candidate is timed
 * over `weights[0..n_weights)` (HOST array of device pointers to the same projection of several layers, visited round-robin so
 * the weights are cold like in the layer loop) and the fastest is kept.  For skinny problems (prompt tail, m = 30) the default
 * pick can be 2x off.  SYNCHRONISES `stream` (the only entry point that does); `out` is scratch. */
int qp_linear_tune(qp_ctx* ctx, const void* x, const void* const* weights, int n_weights, const void* bias, int bias_f32, float alpha,
                   void* out, int64_t m, int64_t n, int64_t k, int act, void* workspace, size_t workspace_bytes, void* stream);
/* The tuner's decision for a problem is per (device, m, n, k, act, bias kind) and PROCESS-wide: the first qp_linear_tune of a problem
 * times the candidates, every later one — from this or any other context of the same device — adopts that pick without timing, and a
 * context that meets the problem in qp_linear_act without having tuned it runs the recorded pick too.  (Two stopwatch runs can disagree
 * on candidates within noise of each other, and a different algorithm is a different fp32 accumulation order: all contexts of a
 * process compute the same bits.)  What is recorded is the ALGORITHM, not its place in a candidate list (the list depends on the
 * workspace limit it was asked with): a context adopts the record only when its own list holds that algorithm within the workspace
 * the caller offers, and re-times otherwise.  qp_linear_plan_choice: returns a qp_status (0.7: the status no longer shares the return
 * value with the index); *choice = index of the heuristic candidate THIS context runs for the problem (0 = hipBLASLt's own first
 * pick), -1 if the context has not seen it; *tuned (optional) = 0 no stopwatch decision on record for the device, 1 on record, 2 on
 * record and this context's plan runs exactly that algorithm.  bias_kind: 0 none, 1 bf16, 2 fp32.  Host-side query, no device work. */
int qp_linear_plan_choice(qp_ctx* ctx, int64_t m, int64_t n, int64_t k, int act, int bias_kind, int* choice, int* tuned);
/* Front end, first step (the reference: HF image processor on the CPU under the GIL, lvu/models/qwen25_lvu_interleaved.py:252-271, 318-340; then
 * H2D of 4 bytes per value, qwen25_lvu.py:691): uint8 frames [n_frames, 3, height, width] on the device -> pixel rows
 * [n_frames / temporal_patch * (height / patch) * (width / patch)][row_elems] bf16 in the HF Qwen2VLImageProcessor order
 * (t, h/merge, w/merge, merge, merge | C, temporal_patch, patch, patch), rescaled + normalised, in one pass.  lut_bf16: 3 x 256 bf16 on the
 * device, lut[c][v] = the normalised value of byte v in channel c — computed by the CALLER with the expression of its own reference path, so
 * the result is bit-identical to it by construction.  row_elems >= 3 * temporal_patch * patch^2 (a multiple of 8); the columns behind the
 * patch are written as zeros (a tile-aligned K for the patch-embedding GEMM: 1176 -> 1280). */
int qp_patchify(qp_ctx* ctx, const void* frames_u8, int n_frames, int height, int width, int patch, int temporal_patch, int merge,
                const void* lut_bf16, void* out, int row_elems, void* stream);
/* out = y * sigmoid(1.702 y) with torch's bf16 rounding steps (hidden_act = quick_gelu). */
int qp_quick_gelu(qp_ctx* ctx, const void* x, void* out, int64_t n_elems, void* stream);

/* ------------------------------------------------------------------------------------------------------------------------------------
 * One segment (a video group, or the prompt tail) through ALL decoder layers in one call: the reference's patched decoder layer
 * (lvu/models/qwen25_lvu.py:122-212) with its attention (:29-120) and prune hook (lvu/utils.py:197-376), looped over the layers the way
 * the group loop does (:671-717) — on the host side this replaces ~13 calls per layer from the caller's language with one call per
 * segment.  It issues exactly the launches the per-operator entry points above issue, in the same order, on `stream`:
 *   x = RMSNorm(h += delta) -> qkv = x Wqkv^T + b -> M-RoPE + K/V append (staging rows + 16-bit norm keys when the layer prunes, the cache
 *   tail otherwise) -> MFMA attention over (cache prefix, new rows) -> o = att Wo^T -> [k smallest norm keys kept: rows staging -> cache tail]
 *   -> x = RMSNorm(h += o) -> act = SiLU(x Wg^T) * (x Wu^T) -> delta = act Wd^T;  after the last layer h += delta.
 * Single-device key-norm path only (prune modes QP_PRUNE_KEY_NORMS_SMALL / QP_PRUNE_KEY_NORMS with a head layout qp_rope_append_keys can
 * fuse); tensor / group-token parallel layers, value-norm and query-score modes and hidden-state pruning stay with the per-operator calls.
 * GEMMs go through qp_linear_act's plans (tune the shapes with qp_linear_tune first); `split_*` cut a projection's rows into two GEMMs
 * ([0, split) and [split, n); 0 = one GEMM), `gate_up_two_gemms` runs gate and up as two [n, I] GEMMs — the decompositions the caller's
 * tuner found fastest for this row count (hipBLASLt's pick is erratic in M).
 * cache_len (HOST, n_layers, in/out): rows in use per layer.  k_keep (HOST, n_layers): tokens the layer keeps, or -1 = append all n.
 * kept_idx (device int32, n_layers rows of kept_idx_stride): layer l's ascending kept list.  attn_events / prune_events: NULL, or
 * 2*n_layers hipEvent_t recorded right before / after each layer's attention / prune launch (measurement).  Nothing is allocated,
 * nothing synchronises. */
typedef struct qp_layer {
  const void* ln1;        /* bf16 [hidden] */
  const void* w_qkv;      /* bf16 [(n_q + 2 n_kv) * 128][hidden], rows: q heads, k heads, v heads */
  const void* b_qkv;      /* bf16 [(n_q + 2 n_kv) * 128] */
  const void* w_o;        /* bf16 [hidden][n_q * 128] */
  const void* ln2;        /* bf16 [hidden] */
  const void* w_gate_up;  /* bf16 [2 * intermediate][hidden], rows: gate then up */
  const void* w_down;     /* bf16 [hidden][intermediate] */
  void* k_cache;          /* bf16 [n_kv][capacity][128] of this layer */
  void* v_cache;
} qp_layer;

typedef struct qp_segment {
  int32_t n_layers, hidden, n_q_heads, n_kv_heads, head_dim, intermediate;
  float rms_eps, attn_scale;
  int64_t n;                    /* rows of this segment */
  int64_t cache_capacity;       /* rows per head of every layer's cache (head stride = capacity * head_dim) */
  int32_t prune_mode;           /* enum qp_prune_mode (key-row modes only) */
  int32_t attend_prefix;        /* 0: a video group under adaptive_local_attention = False (qwen25_lvu.py:700-714): no cache prefix */
  int64_t split_qkv, split_o, split_gate_up, split_down;
  int32_t gate_up_two_gemms, reserved_;
  void *h, *x, *qkv, *q, *att, *o, *gate_up, *act, *down;     /* bf16 [n][...] activations, caller-owned */
  void *k_stage, *v_stage;      /* bf16 [n_kv][n][128]: the segment's new K/V while the layer prunes */
  uint16_t* norm_keys;          /* [n] */
  int32_t* kept_idx;
  int64_t kept_idx_stride;
  const void *cos, *sin;        /* M-RoPE table of the segment (qp_mrope_table) */
  void* attn_ws;  size_t attn_ws_bytes;
  void* gemm_ws;  size_t gemm_ws_bytes;
  void** attn_events;
  void** prune_events;          /* same, around each pruning layer's qp_prune_keys launch (entries of non-pruning layers are not recorded) */
} qp_segment;

int qp_prefill_segment(qp_ctx* ctx, const qp_segment* seg, const qp_layer* layers, int64_t* cache_len, const int64_t* k_keep,
                       void* stream);

/* All blocks of the Qwen2-VL vision tower (transformers Qwen2VLVisionBlock [3P]) in one call — the tower's counterpart of
 * qp_prefill_segment: per block  y = LayerNorm(x += pending) -> qkv = y Wqkv^T + b -> 2-D rotary -> full attention inside each temporal
 * patch -> pending = a Wp^T + b -> y = LayerNorm(x += pending) -> z = Swish(1.702 (y W1^T + b1)) -> pending = z W2^T / 1.702 + b2
 * (fc1 + quick-GELU as ONE GEMM, see qp_linear_act), the same launches quickvideo_amd/vit.py issues one by one.  On return x holds the
 * residual stream WITHOUT the last block's MLP output, which is left in `pending` (the caller's final LayerNorm adds it: qp_add_layernorm).
 * x, y, pending bf16 [n][dim]; qkv bf16 [n][3*dim]; att bf16 [n][dim]; z bf16 [n][mlp_dim]; cos/sin fp32 [n][head_dim/2];
 * fc1_bias_scaled fp32 [mlp_dim] = 1.702 * b1.  n = n_seq * seq_len.  GEMM plans as for qp_linear_act (tune first). */
typedef struct qp_vit_block {
  const void *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc1_w, *fc1_bias_scaled, *fc2_w, *fc2_b;
} qp_vit_block;

int qp_vit_blocks(qp_ctx* ctx, const qp_vit_block* blocks, int n_blocks, int64_t n_seq, int64_t seq_len, int dim, int heads, int mlp_dim,
                  void* x, void* y, void* qkv, void* att, void* pending, void* z, const float* cos, const float* sin, float ln_eps,
                  void* gemm_ws, size_t gemm_ws_bytes, void* stream);

/* ---- frame ring of the overlap producer (reference: the daemon thread + Queue(maxsize=3) + 10 ms polling of
 *      lvu/models/qwen25_lvu_interleaved.py:237-342, 853-871) -------------------------------------------------------------------------
 * A NATIVE producer thread (std::thread of this library) pulls frame groups from a source callback straight into pinned host slots,
 * enqueues each H2D copy on `copy_stream` and signals the consumer with events; nothing polls.  Slot reuse is ordered in both
 * directions: a host slot is refilled only after the previous copy out of it has finished (the producer thread waits on the event),
 * a device slot is overwritten only after the consumer's last GPU read of it (the copy stream waits on the event recorded by
 * qp_frame_ring_mark_read / _release).  The frame source runs one group ahead of the ring (it fills host slot g % depth while the
 * device slots still hold groups g-depth .. g-1), like the reference's thread that decodes before it blocks in put().
 * Ownership: host slots (pinned) and device slots are the CALLER's (depth buffers of slot_bytes each) and must outlive the ring;
 * the ring owns its events and its thread.  ctx == NULL and dev_slots == NULL: host-only ring (no HIP call; acquire hands out the
 * host slot, the source is not called for a slot before its group has been released).
 * One video per ring: create -> start[_file] -> acquire / mark_read / release per group, in order -> stop -> destroy.
 * qp_frame_ring_acquire and _stop BLOCK on the producer thread: call them with the GIL released (ctypes.CDLL does) — a Python source
 * callback needs the GIL to run. */
typedef struct qp_frame_ring qp_frame_ring;
/* Fill `dst` (capacity bytes) with frame group g (uint8 [frames, 3, H, W], contiguous); return the bytes written, 0 when the video has
 * no group g (early end), < 0 on failure.  Called on the ring's thread, one group at a time, g ascending from 0. */
typedef int64_t (*qp_frame_source_fn)(void* user, int64_t g, void* dst, size_t capacity);
int qp_frame_ring_create(qp_ctx* ctx, int depth, size_t slot_bytes, void* const* host_slots, void* const* dev_slots, void* copy_stream,
                         qp_frame_ring** out);
int qp_frame_ring_start(qp_frame_ring* ring, qp_frame_source_fn source, void* user, int64_t n_groups);
/* Built-in source for pre-decoded videos (a raw uint8 [F, 3, H, W] array in a file, e.g. the data section of a .npy): group g =
 * frames frame_idx[g*frames_per_group ...], each pread() at data_offset + idx*frame_bytes directly into the pinned slot by up to
 * io_threads threads.  No interpreter anywhere on the frame path. */
int qp_frame_ring_start_file(qp_frame_ring* ring, const char* path, int64_t data_offset, int64_t frame_bytes, const int64_t* frame_idx,
                             int64_t n_frames, int frames_per_group, int io_threads);
/* origin_event: a hipEvent_t (timing enabled) the caller recorded; qp_frame_ring_h2d_ms reports against it.  Optional. */
int qp_frame_ring_set_origin(qp_frame_ring* ring, void* origin_event);
/* Blocks the HOST until group g's copy has been enqueued, then makes `consumer_stream` wait for it ON THE DEVICE (no host sync on the
 * copy).  *ptr_out = the device slot (host slot of a host-only ring), *bytes_out = what the source wrote.  Errors of the source or of
 * the producer thread surface here. */
int qp_frame_ring_acquire(qp_frame_ring* ring, int64_t g, void* consumer_stream, void** ptr_out, size_t* bytes_out);
/* The same with a bounded host wait: returns QP_ERR_TIMEOUT (qp_last_error untouched, ring state untouched) when group g has not been
 * published within timeout_ms; timeout_ms < 0 waits for good.  A host that must stay interruptible — the reference's consumer polls
 * its queue every 10 ms (qwen25_lvu_interleaved.py:853-871), so Ctrl-C works while the decoder hangs — calls this in a loop. */
int qp_frame_ring_acquire_for(qp_frame_ring* ring, int64_t g, void* consumer_stream, int64_t timeout_ms, void** ptr_out, size_t* bytes_out);
/* Record "last GPU read of group g's slot" on consumer_stream now (optional; _release records it if nobody did). */
int qp_frame_ring_mark_read(qp_frame_ring* ring, int64_t g, void* consumer_stream);
int qp_frame_ring_release(qp_frame_ring* ring, int64_t g, void* consumer_stream);
int qp_frame_ring_stop(qp_frame_ring* ring);                       /* cancel + join; safe at any point, idempotent */
/* out[0..5] = seconds inside the source / waiting for a released slot / waiting for a previous copy / enqueueing copies, groups
 * published, groups the video has (shrinks if the source ended early). */
int qp_frame_ring_stats(qp_frame_ring* ring, double* out, int n_out);
/* Per group: when its H2D copy finished, in ms after the origin event (NaN: unknown / no origin).  After the producer has ended. */
int qp_frame_ring_h2d_ms(qp_frame_ring* ring, float* out, int64_t n_out);
void qp_frame_ring_destroy(qp_frame_ring* ring);

#ifdef __cplusplus
}
#endif
#endif /* QUICKPREFILL_H_ */
