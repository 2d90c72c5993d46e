// Shared declarations of libquickprefill (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/quickprefill.h"

struct qp_ctx {
  int device;
  int cus;
  int lds_per_cu;
  void* lt;          // hipBLASLt handles (one per stream) and GEMM plans of THIS context/device (qp_linear.hip); freed by qp_destroy
};
void qp_lt_destroy(void* lt_state);

// Developer A/B switches of the launch paths.  NOT read from the environment at launch time (rounds 1-4 called getenv() in every
// attention launch): the table is filled ONCE, when the library is first used, from the QP_* variables named below, and changed at run
// time only through qp_dev_switch() (include/quickprefill.h) — the launch paths read relaxed atomics.
#include <atomic>
struct qp_dev_switches {
  std::atomic<int> attn_variant{0};        // QP_ATTN_VARIANT: 0 production; 1 v1, 2 no kv split, 3 no XCD map, 4 s4, 7/8 4-/8-wave s6 (9, 10: EXPERIMENTS builds)
  std::atomic<int> attn_force_split{0};    // QP_ATTN_FORCE_SPLIT: every work item cut into this many kv ranges (0 = planner's choice)
  std::atomic<int> s6_prio{0};             // QP_S6_PRIO
  std::atomic<int> s6_early_out{3};        // QP_S6_EARLY_OUT: bit 0 rows past the last query, bit 1 causal diagonal
  std::atomic<int> decode_attn_valu{0};    // QP_DECODE_ATTN=valu: first (VALU) form of the single-query decode attention
  std::atomic<int> attn_debug{0};          // QP_ATTN_DEBUG: print every new attention plan
  std::atomic<int> attn_flat{-1};          // QP_ATTN_FLAT: flat (stream-K) split of the attention launch: -1 by cost (default), 0 never, 1 whenever eligible
};
qp_dev_switches& qp_dev();

// prune_mode argument of the seam-1 entry points (include/quickprefill.h: enum qp_prune_mode): bit 0 = keep the k LARGEST norms,
// bit 1 = score the VALUE rows
static inline int qp_mode_largest(int prune_mode) { return prune_mode & 1; }
static inline int qp_mode_values(int prune_mode) { return (prune_mode >> 1) & 1; }

// thread-local error message (qp_api.cpp)
int qp_fail(int status, const char* fmt, ...);
int qp_check_launch(const char* what);

#define QP_REQUIRE(cond, status, ...) do { if (!(cond)) return qp_fail((status), __VA_ARGS__); } while (0)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
// round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet): gfx950's own conversion instruction (v_cvt_pk_bf16_f32 — what the
// `(__bf16)` cast compiles to).  Rounds 1-2 carried a six-instruction integer emulation here; tools/probe/probe_cvt_bf16.hip compares
// the two over ALL 2^32 fp32 patterns on the device: 0 mismatches, NaN payloads included — so every bit-exact claim is untouched and
// the RoPE / RMSNorm / SwiGLU kernels lose ~100 VALU instructions per 16-byte chunk.
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }
// Correctly rounded fp32 square root.  hipcc's sqrtf/__fsqrt_rn can be 1 ulp off (measured: s = 262.03513 gave
// 16.187498 instead of 16.1875, which flipped a bf16 round-to-even tie and with it a kept index); the fp64 square root
// rounded once to fp32 is exact because sqrt of an fp32 value is never within 2^-48 of an fp32 rounding boundary.
__device__ __forceinline__ float sqrt_rn_f32(float s) {
  float r = (float)sqrt((double)s);
  // opaque to the optimiser: a following `(__bf16)r` must round THIS fp32 value.  Without the barrier LLVM folds
  // fptrunc(fptrunc(double)) into one double -> bf16 rounding, which differs from fp32 -> bf16 whenever the fp32 value sits exactly on
  // a bf16 tie (1 norm in ~300 000: caught by test_select_edge_sizes[300000-60000-2] the moment f32_to_bf16_bits became a cast).
  asm volatile("" : "+v"(r));
  return r;
}

// Kernels that need more than 64 KB of dynamic LDS opt in with hipFuncSetAttribute — per DEVICE (the attribute lives in the
// per-device function object), so the guard is a bit per device ordinal, not one flag per process.
static inline int qp_opt_in_lds(std::atomic<unsigned long long>& done, const void* fn, int bytes, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_relaxed) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return qp_fail(QP_ERR_HIP, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
  done.fetch_or(bit, std::memory_order_relaxed);
  return 0;
}

// kernel launchers (one per .hip file)
int qp_launch_mrope_table(const int64_t* pos, int64_t n, const int32_t* sections, float theta, int head_dim, void* cos_out,
                          void* sin_out, hipStream_t s);
int qp_launch_rope_append(const void* qkv, const void* cos, const void* sin, int64_t n, int hq, int hkv, void* q_out,
                          void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0, float* head_sumsq,
                          uint16_t* norm_keys, int largest, hipStream_t s);
bool qp_rope_can_fuse_keys(int hq, int hkv);
int qp_launch_norm_keys(const float* head_sumsq, int n_heads, int64_t n, uint16_t* norm_keys, int largest, hipStream_t s);
int qp_launch_query_scores(const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int hq, int hkv,
                           const float* value_sumsq, uint16_t* keys_out, uint16_t* scores_out, void* workspace, hipStream_t s);
int qp_launch_query_head_sums(const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int hq, int hkv,
                              uint16_t* head_sums_out, void* workspace, hipStream_t s);
int qp_launch_query_scores_final(const uint16_t* head_sums, int hq_total, int64_t n, const float* value_sumsq, int hkv_total, uint16_t* keys_out,
                                 uint16_t* scores_out, hipStream_t s);
int qp_launch_prune_keys(const uint16_t* norm_keys, int64_t n, int64_t k, const void* k_src, const void* v_src, int64_t src_head_stride,
                         int hkv, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0, int32_t* kept, hipStream_t s);
int qp_launch_key_sumsq(const void* k, int64_t head_stride, int64_t row0, int64_t n, int hkv, float* head_sumsq,
                        hipStream_t s);
int qp_launch_select(const float* head_sumsq, int n_heads, int64_t n, int64_t k, int32_t* kept, uint16_t* norm_bits,
                     void* ws, int largest, hipStream_t s, const uint16_t* keys_in = nullptr);
int qp_launch_gather_kv(const void* k_src, const void* v_src, int64_t src_head_stride, const int32_t* idx, int64_t k,
                        int hkv, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0, hipStream_t s);
int qp_launch_gather_rows(const void* src, const int32_t* idx, int64_t k, int64_t row_bytes, void* dst, hipStream_t s);
int qp_launch_sp_unpack(const void* gathered, int world, int hkv, int64_t m2, int64_t n, void* k_stage, void* v_stage,
                        int64_t stage_head_stride, float* sumsq_out, hipStream_t s);
int qp_launch_copy_rows_kv(const void* k_src, const void* v_src, int64_t src_head_stride, int64_t k, int hkv, void* k_dst,
                           void* v_dst, int64_t dst_head_stride, int64_t dst_row0, hipStream_t s);
int qp_launch_add_rmsnorm(void* h, const void* delta, const void* w, void* out, int64_t n, int hidden, float eps,
                          hipStream_t s);
int qp_launch_add_inplace(void* h, const void* delta, int64_t n_elems, hipStream_t s);
int qp_launch_swiglu(const void* gate, const void* up, int64_t row_elems, int64_t n, int inter, void* out, hipStream_t s);
int qp_launch_prefill_attn(const qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix,
                           int64_t prefix_head_stride, int64_t prefix_len, const void* k_new, const void* v_new,
                           int64_t new_head_stride, int64_t n, int64_t q_row0, int64_t nq, int hq, int hkv, float scale, void* out,
                           void* workspace, size_t workspace_bytes, hipStream_t s);
size_t qp_attn_workspace_bytes_impl(const qp_ctx* ctx, int64_t n, int64_t prefix_len, int hq, int hkv);
int qp_launch_vit_attn(const qp_ctx* ctx, const void* qkv, int64_t n_seq, int64_t S, int heads, float scale, void* out,
                       const int* cu_seqlens, hipStream_t s);
int qp_launch_vit_rope(void* qkv, const float* cos_t, const float* sin_t, int64_t n, int heads, int head_dim, hipStream_t s);
int qp_launch_quick_gelu(const void* x, void* out, int64_t n_elems, hipStream_t s);
int qp_launch_patchify(const void* frames, const void* lut, void* out, int n_frames, int H, int W, int ps, int tp, int mg, int row_elems, hipStream_t s);
int qp_launch_add_layernorm(void* x, const void* delta, const void* w, const void* b, void* out, int64_t n, int hidden, float eps,
                            hipStream_t s);
int qp_launch_gemv(const qp_ctx* ctx, const void* w, const void* x, const void* norm_w, float eps, const void* bias, void* out,
                   int64_t n_out, int64_t k, int mode, hipStream_t s);
int qp_launch_decode_rope(const void* qkv, const int64_t* state, const void* cos_t, const void* sin_t, float theta, int hq, int hkv,
                          void* q_out, void* k_cache, void* v_cache, int64_t head_stride, hipStream_t s);
size_t qp_decode_attn_workspace_bytes_impl(const qp_ctx* ctx, int hq, int hkv);
int qp_launch_decode_attn(const qp_ctx* ctx, const void* q, const void* k_cache, const void* v_cache, int64_t head_stride,
                          const int64_t* state, int hq, int hkv, float scale, void* out, void* workspace, hipStream_t s);
int qp_launch_decode_attn_fused(const qp_ctx* ctx, const void* qkv, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                                int64_t head_stride, const int64_t* state, int hq, int hkv, float scale, void* out, void* workspace,
                                hipStream_t s);
int qp_launch_decode_advance(int64_t* state, int n, hipStream_t s);
int qp_launch_tail_keys(const void* rows, int64_t head_stride, int64_t row0, int64_t n, int hkv, uint16_t* norm_keys, int largest,
                        int* sync_words, int n_sync_words, hipStream_t s);
int qp_prune_tail_inplace_capacity(int cus);
int qp_launch_prune_tail_inplace(const uint16_t* norm_keys, int64_t n, int64_t k, void* k_cache, void* v_cache, int64_t head_stride,
                                 int64_t past_len, int hkv, int32_t* kept, int* sync_words, hipStream_t s);
int qp_launch_linear_tune(qp_ctx* ctx, const void* x, const void* const* ws_list, int n_ws, const void* bias, int bias_f32, float alpha, void* out,
                          int64_t m, int64_t n, int64_t k, int act, void* workspace, size_t workspace_bytes, hipStream_t s, int* chosen);
int qp_linear_plan_choice_impl(qp_ctx* ctx, int64_t m, int64_t n, int64_t k, int act, int bias_kind, int* choice, int* tuned);
int qp_launch_linear_act(qp_ctx* ctx, const void* x, const void* w, const void* bias, int bias_f32, float alpha, void* out, int64_t m, int64_t n,
                         int64_t k, int act, void* workspace, size_t workspace_bytes, hipStream_t s);
