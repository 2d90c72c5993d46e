// Query-attention-score pruning: the scoring of LVUCache.update in query-based mode, lvu/lvu_cache.py:97-117.
//   prompt queries q [m][Hq][128] (RoPE'd) x the group's own keys k [Hkv][n][128] (RoPE'd, NOT the past):
//     a = bf16(q.k)           (fp32 accumulate, einsum output in the model dtype)
//     b = bf16(a / sqrt(128))
//     p = bf16(softmax_fp32(b) over the n keys)
//     s1[h][t] = bf16(sum over the m queries of p)        s[t] = bf16(mean over the Hq heads of s1)
//   and the sort key handed to qp_prune_keys: the complemented bf16 pattern of s (k LARGEST scores, ties -> lowest index:
//   utils.py:55-57), or of bf16(s * ||v_t||) for query_attention_weights_by_value_norm (utils.py:58-62).
// m is a few dozen tokens and the whole thing is ~1 GFLOP per layer: VALU dot products, two small kernels, no MFMA.
// This mode is the README's "Attention Scores" baseline (worse accuracy/throughput trade than key norms) — correctness first.
#include "qp_common.h"

__device__ __forceinline__ float block_reduce_256(float v, bool is_max, float* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, t) : v + t;
  }
  __syncthreads();                       // sh may still be read from the previous reduction
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

// grid (m, Hq); dynamic LDS: n floats.  P[(h*m + qi)*n + t] = bf16 probability.
__global__ __launch_bounds__(256) void qscore_prob_kernel(const uint4* __restrict__ q, const uint4* __restrict__ k, int64_t k_hs16,
                                                          int n, int m, int hq, int hkv, float inv_sqrt_d_div,
                                                          uint16_t* __restrict__ P) {
  extern __shared__ float row[];
  __shared__ float red[4];
  const int qi = blockIdx.x, h = blockIdx.y, c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int kvh = h / (hq / hkv);
  const uint4 qv = q[((int64_t)qi * hq + h) * 16 + c];
  const unsigned qw[4] = {qv.x, qv.y, qv.z, qv.w};
  float qf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = bf16_bits_to_f32((unsigned short)(qw[e >> 1] >> ((e & 1) * 16)));
  for (int t = grp; t < n; t += 16) {
    const uint4 kv = k[(int64_t)kvh * k_hs16 + (int64_t)t * 16 + c];
    const unsigned kw[4] = {kv.x, kv.y, kv.z, kv.w};
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = __builtin_fmaf(qf[e], bf16_bits_to_f32((unsigned short)(kw[e >> 1] >> ((e & 1) * 16))), s);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s = s + __shfl_xor(s, o, 16);
    if (c == 0) row[t] = round_bf16(round_bf16(s) / inv_sqrt_d_div);      // bf16(q.k) / sqrt(D), rounded to bf16 again
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = threadIdx.x; t < n; t += 256) mx = fmaxf(mx, row[t]);
  mx = block_reduce_256(mx, true, red);
  float sum = 0.f;
  for (int t = threadIdx.x; t < n; t += 256) { const float e = expf(row[t] - mx); row[t] = e; sum += e; }
  sum = block_reduce_256(sum, false, red);
  uint16_t* out = P + ((int64_t)h * m + qi) * n;
  for (int t = threadIdx.x; t < n; t += 256) out[t] = f32_to_bf16_bits(row[t] / sum);
}

// thread per (head, key): s1[h][t] = bf16(sum over the m prompt queries of the bf16 probabilities)
__global__ __launch_bounds__(256) void qscore_headsum_kernel(const uint16_t* __restrict__ P, int n, int m, int hq, uint16_t* __restrict__ s1) {
  const int t = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (t >= n) return;
  float acc = 0.f;
  const uint16_t* p = P + (int64_t)h * m * n + t;
  for (int qi = 0; qi < m; ++qi) acc += bf16_bits_to_f32(p[(int64_t)qi * n]);
  s1[(int64_t)h * n + t] = f32_to_bf16_bits(acc);
}

// thread per key t: mean over ALL heads of the per-head sums (ascending head order, bf16 result), optional value-norm weighting,
// complemented key out.  Tensor parallelism all-gathers the ranks' s1 blocks (rank-major = ascending head order) in front of this.
__global__ __launch_bounds__(256) void qscore_final_kernel(const uint16_t* __restrict__ s1, int n, int hq, const float* __restrict__ value_sumsq,
                                                           int hkv, uint16_t* __restrict__ keys_out, uint16_t* __restrict__ scores_out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float acc = 0.f;
  for (int h = 0; h < hq; ++h) acc += bf16_bits_to_f32(s1[(int64_t)h * n + t]);
  float sc = round_bf16(acc / (float)hq);
  if (scores_out) scores_out[t] = f32_to_bf16_bits(sc);
  if (value_sumsq) {                                   // * ||v_t|| over all kv heads (bf16 norm, heads added in ascending order)
    float s = value_sumsq[t];
    for (int h = 1; h < hkv; ++h) s = s + value_sumsq[(int64_t)h * n + t];
    sc = round_bf16(sc * round_bf16(sqrt_rn_f32(s)));
  }
  keys_out[t] = (uint16_t)~f32_to_bf16_bits(sc);
}

static int launch_probs(const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int hq, int hkv, void* workspace,
                        hipStream_t s) {
  static std::atomic<unsigned long long> lds_ok{0};
  const size_t smem = (size_t)n * 4;
  if (smem > 48 * 1024)
    if (int rc = qp_opt_in_lds(lds_ok, (const void*)qscore_prob_kernel, 160 * 1024 - 256, "qscore_prob")) return rc;
  qscore_prob_kernel<<<dim3((unsigned)m, (unsigned)hq), 256, smem, s>>>((const uint4*)q_prompt, (const uint4*)k_group, k_head_stride / 8,
                                                                        (int)n, (int)m, hq, hkv, sqrtf(128.0f), (uint16_t*)workspace);
  return qp_check_launch("qscore_prob");
}

// workspace = [P: hq*m*n bf16 | pad to 256 B | s1: hq*n bf16]
static size_t probs_bytes(int64_t n, int64_t m, int hq) { return ((size_t)hq * m * n * 2 + 255) & ~(size_t)255; }

int qp_launch_query_head_sums(const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int hq, int hkv,
                              uint16_t* head_sums_out, void* workspace, hipStream_t s) {
  if (int rc = launch_probs(q_prompt, k_group, k_head_stride, n, m, hq, hkv, workspace, s)) return rc;
  qscore_headsum_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)hq), 256, 0, s>>>((const uint16_t*)workspace, (int)n, (int)m, hq, head_sums_out);
  return qp_check_launch("qscore_headsum");
}

int qp_launch_query_scores_final(const uint16_t* head_sums, int hq_total, int64_t n, const float* value_sumsq, int hkv_total, uint16_t* keys_out,
                                 uint16_t* scores_out, hipStream_t s) {
  qscore_final_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(head_sums, (int)n, hq_total, value_sumsq, hkv_total, keys_out, scores_out);
  return qp_check_launch("qscore_final");
}

int qp_launch_query_scores(const void* q_prompt, const void* k_group, int64_t k_head_stride, int64_t n, int64_t m, int hq, int hkv,
                           const float* value_sumsq, uint16_t* keys_out, uint16_t* scores_out, void* workspace, hipStream_t s) {
  uint16_t* s1 = (uint16_t*)((char*)workspace + probs_bytes(n, m, hq));
  if (int rc = qp_launch_query_head_sums(q_prompt, k_group, k_head_stride, n, m, hq, hkv, s1, workspace, s)) return rc;
  return qp_launch_query_scores_final(s1, hq, n, value_sumsq, hkv, keys_out, scores_out, s);
}
