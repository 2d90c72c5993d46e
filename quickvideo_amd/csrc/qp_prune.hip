// Seam 1 kernels: canonical key sum-of-squares, k-smallest select, KV gather / compaction.
// Reference semantics: lvu/utils.py:133-136 (key_norms_small), :190-194 (mask), :266-342 (gather + cat).
// All HBM-bound byte/integer work: coalesced 16-B accesses, 16 lanes per 256-B head row.
#include "qp_common.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// K4: per-head sum of squares in the canonical order (oracle: key_sumsq_heads).
//   16 lanes per head row (D=128): lane c owns elements 8c..8c+7, accumulates them left to right
//   (x*x is exact in fp32), then an xor butterfly 1,2,4,8 over the 16 lanes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float chunk_sumsq(uint4 v) {
  unsigned w[4] = {v.x, v.y, v.z, v.w};
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
    s = __builtin_fmaf(lo, lo, s);
    s = __builtin_fmaf(hi, hi, s);
  }
  return s;
}

__device__ __forceinline__ float row16_butterfly(float s) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) s = s + __shfl_xor(s, m, 16);
  return s;
}

__global__ __launch_bounds__(256) void key_sumsq_kernel(const uint4* __restrict__ k, int64_t head_stride16, int64_t row0,
                                                        int64_t n, int hkv, float* __restrict__ head_sumsq) {
  const int c = threadIdx.x & 15;
  const int64_t rows = n * hkv;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
    const int64_t h = r / n, t = r - h * n;
    uint4 v = k[h * head_stride16 + (row0 + t) * 16 + c];
    float s = row16_butterfly(chunk_sumsq(v));
    if (c == 0) head_sumsq[h * n + t] = s;
  }
}

int qp_launch_key_sumsq(const void* k, int64_t head_stride, int64_t row0, int64_t n, int hkv, float* head_sumsq,
                        hipStream_t s) {
  int64_t rows = n * hkv;
  int grid = (int)((rows + 15) / 16 < 4096 ? (rows + 15) / 16 : 4096);
  if (grid < 1) grid = 1;
  key_sumsq_kernel<<<grid, 256, 0, s>>>((const uint4*)k, head_stride / 8, row0, n, hkv, head_sumsq);
  return qp_check_launch("key_sumsq");
}

// ------------------------------------------------------------------------------------------------
// K5: k-smallest select on bf16 norms, ties -> lowest index, ascending index list out.
//   One 1024-thread workgroup (n <= 65536): norms (16-bit patterns) live in LDS;
//   two 256-bin histogram passes find the threshold pattern tau, then an ordered two-scan compaction
//   emits  {t : key<tau}  U  {first r ties of key==tau}.  No host round trip (the reference does
//   .tolist() + torch.tensor + nonzero().cpu(): utils.py:136,191,284).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned block_excl_scan_1024(unsigned v, unsigned* wave_tot, unsigned* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    unsigned w = lane < 16 ? wave_tot[lane] : 0u, wi = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      unsigned t = __shfl_up(wi, o, 64);
      if (lane >= o) wi += t;
    }
    if (lane < 16) wave_tot[lane] = wi - w;   // exclusive wave offsets
    if (lane == 15) *total = wi;
  }
  __syncthreads();
  unsigned r = wave_tot[wave] + incl - v;
  __syncthreads();
  return r;
}

// finds bucket b with cum_before(b) < kk <= cum_before(b)+hist[b]; result in res[0]=b, res[1]=cum_before(b).
// hist = `nh` partial histograms of 256 bins with stride `hstride` (summed here).  One wave does it: lane l owns the
// four consecutive bins 4l..4l+3 and a shuffle scan orders the lanes -> a single barrier instead of a 16-barrier
// block-wide Hillis-Steele scan (the select is latency-bound: one workgroup, a chain of dependent steps).
__device__ __forceinline__ void find_bucket_256(const unsigned* hist, int nh, int hstride, unsigned kk, unsigned* res) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    unsigned v[4] = {0, 0, 0, 0};
    for (int w = 0; w < nh; ++w) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += hist[w * hstride + 4 * lane + i];
    }
    const unsigned s4 = v[0] + v[1] + v[2] + v[3];
    unsigned incl = s4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      unsigned t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    unsigned before = incl - s4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (before < kk && kk <= before + v[i]) { res[0] = (unsigned)(4 * lane + i); res[1] = before; }
      before += v[i];
    }
  }
  __syncthreads();
}

// kLds: norm patterns live in LDS (n <= 65536); otherwise in a caller-provided global scratch (any n: the one workgroup
// then streams them from L2 a few times — the rarely used "single group" baseline mode of very long videos).
template <bool kLds>
__global__ __launch_bounds__(1024) void select_kernel(const float* __restrict__ head_sumsq, int n_heads, int n, int k,
                                                      int32_t* __restrict__ kept, uint16_t* __restrict__ norm_bits_out,
                                                      uint16_t* __restrict__ keys_glb, int largest, const uint16_t* __restrict__ keys_in) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* hist = (unsigned*)smem;            // 256
  unsigned* scan = hist + 256;                 // 256
  unsigned* wave_tot = scan + 256;             // 16
  unsigned* res = wave_tot + 16;               // 4
  unsigned* tot = res + 4;                     // 4
  uint16_t* keys = kLds ? (uint16_t*)(tot + 4) : keys_glb;       // n
  const int tid = threadIdx.x;

  if (keys_in) {                                   // ready-made 16-bit sort keys (qp_select_keys: norm keys, complemented query scores)
    if (kLds) for (int t = tid; t < n; t += 1024) keys[t] = keys_in[t];
    else keys = const_cast<uint16_t*>(keys_in);    // large n: stream them from where they are
  } else {
    for (int t = tid; t < n; t += 1024) {
      float s = head_sumsq[t];
      for (int h = 1; h < n_heads; ++h) s = s + head_sumsq[(int64_t)h * n + t];
      uint16_t b = f32_to_bf16_bits(sqrt_rn_f32(s));
      keys[t] = largest ? (uint16_t)~b : b;        // k largest == k smallest of the complemented pattern (ties: lowest index)
      if (norm_bits_out) norm_bits_out[t] = b;
    }
  }
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int t = tid; t < n; t += 1024) atomicAdd(&hist[keys[t] >> 8], 1u);
  __syncthreads();
  find_bucket_256(hist, 1, 256, (unsigned)k, res);
  const unsigned b1 = res[0], c1 = res[1];
  __syncthreads();
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  for (int t = tid; t < n; t += 1024) { unsigned key = keys[t]; if ((key >> 8) == b1) atomicAdd(&hist[key & 255u], 1u); }
  __syncthreads();
  find_bucket_256(hist, 1, 256, (unsigned)k - c1, res);
  const unsigned tau = (b1 << 8) | res[0];
  const unsigned n_less = c1 + res[1];
  const unsigned r_ties = (unsigned)k - n_less;      // >= 1 ties (key == tau) to take, lowest index first
  __syncthreads();

  const int chunk = (n + 1023) / 1024;
  const int t0 = tid * chunk, t1 = min(n, t0 + chunk);
  unsigned lt = 0, eq = 0;
  for (int t = t0; t < t1; ++t) { unsigned key = keys[t]; lt += key < tau; eq += key == tau; }
  const unsigned eq_before = block_excl_scan_1024(eq, wave_tot, tot);
  unsigned take = 0;
  if (eq_before < r_ties) take = min(eq, r_ties - eq_before);
  const unsigned pos0 = block_excl_scan_1024(lt + take, wave_tot, tot);
  unsigned pos = pos0, taken = 0;
  for (int t = t0; t < t1; ++t) {
    unsigned key = keys[t];
    bool keep = key < tau;
    if (key == tau && taken < take) { keep = true; ++taken; }
    if (keep) kept[pos++] = t;
  }
}


// ------------------------------------------------------------------------------------------------
// K5+K6, the engine's path since round 2: ONE launch, no 150 KB of LDS, no fp64 square roots per workgroup.
//   * The 16-bit norm key of every token (bf16 pattern of the cross-head norm; complemented for "k largest") is produced once,
//     by the RoPE/append kernel that already holds the key rows (qp_rope.hip) — or by norm_keys_kernel below when the per-head
//     sums had to cross ranks first (tensor / group-token parallel) or come from the value rows.
//   * prune_keys_kernel: one 256-thread workgroup per 16 consecutive tokens.  Every workgroup holds ALL n keys in registers
//     (2 bytes per token from L2: n <= 8192) and finds the threshold key tau by a two-pass 256-bin radix select in LDS
//     (8 bank-staggered histogram copies: norms concentrate on a few dozen values, so same-address adds are 8-way, not 64-way),
//     counts the kept tokens in front of its slice from the same registers, and moves the K/V rows of its own kept tokens
//     staging -> arena (16 lanes x 16 B per 256-B row, all loads in flight before the first store).  Tie rule and order as
//     select_kernel: keys below tau, then the first (k - n_less) ties in index order; ascending index list out.
//   Tried first (as the round-1 review suggested): a global two-level histogram filled with atomics from the RoPE kernel's
//   epilogue and read by a scan+gather kernel.  The scan+gather kernel ran in 8.4-9.3 us back to back, but the atomics did not
//   pay: a few dozen hot bins serialise at L2 round-trip latency — +43 us per RoPE launch at n=5760 (8.47 vs 3.68 ms per cfg2
//   pass), 65 us for a stand-alone histogram kernel.  (profiles/r2_prune_hist_atomics.txt)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_keys_kernel(const float* __restrict__ head_sumsq, int n_heads, int n,
                                                        uint16_t* __restrict__ norm_keys, int largest) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float s = head_sumsq[t];
  for (int h = 1; h < n_heads; ++h) s = s + head_sumsq[(int64_t)h * n + t];
  uint16_t b = f32_to_bf16_bits(sqrt_rn_f32(s));
  norm_keys[t] = largest ? (uint16_t)~b : b;
}

int qp_launch_norm_keys(const float* head_sumsq, int n_heads, int64_t n, uint16_t* norm_keys, int largest, hipStream_t s) {
  norm_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(head_sumsq, n_heads, (int)n, norm_keys, largest);
  return qp_check_launch("norm_keys");
}

// exclusive scan of one value per thread over the 256 threads of the workgroup (one barrier inside)
__device__ __forceinline__ unsigned block_excl_scan_256(unsigned v, unsigned* wave_tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned before = 0;
#pragma unroll
  for (int w = 0; w < 3; ++w) before += (w < wave) ? wave_tot[w] : 0u;
  return before + incl - v;
}

#define QP_PRUNE_TS 16       // tokens per workgroup
#define QP_HIST_COPIES 8
#define QP_HIST_STRIDE 264   // 256 bins + 8: copy c of bin b sits in bank (b + 8c) % 64

#define QP_PRUNE_MAX_N 8192  // keys of the whole group in LDS (16 KB)

__global__ __launch_bounds__(256) void prune_keys_kernel(const uint16_t* __restrict__ keys_g, int n, int k,
                                                         const uint4* __restrict__ k_src, const uint4* __restrict__ v_src,
                                                         int64_t src_hs16, int hkv, uint4* __restrict__ k_dst,
                                                         uint4* __restrict__ v_dst, int64_t dst_hs16, int64_t dst_row0,
                                                         int32_t* __restrict__ kept) {
  __shared__ __attribute__((aligned(16))) uint16_t keys[QP_PRUNE_MAX_N];
  __shared__ unsigned hist[2][QP_HIST_COPIES * QP_HIST_STRIDE];
  __shared__ unsigned wave_tot[2][4];
  __shared__ unsigned res[4];                  // b1, count before b1, low byte, count before tau inside b1
  __shared__ unsigned red[2][4];
  __shared__ int s_tok[QP_PRUNE_TS], s_pos[QP_PRUNE_TS];
  __shared__ int s_nk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * QP_PRUNE_TS;
  // the code is kept small on purpose (rolled loops, keys in LDS): between two launches of this kernel the GEMMs evict it from
  // the instruction cache, and every cold line of code is a serial miss in a kernel that is nothing but a latency chain

  // all n keys -> LDS, 8 per 16-byte load, the last n % 8 one by one
  const int n8 = n >> 3;
  for (int i = tid; i < n8; i += 256) ((uint4*)keys)[i] = ((const uint4*)keys_g)[i];
  if (tid < (n & 7)) keys[n8 * 8 + tid] = keys_g[n8 * 8 + tid];
  for (int i = tid; i < 2 * QP_HIST_COPIES * QP_HIST_STRIDE; i += 256) (&hist[0][0])[i] = 0u;
  __syncthreads();
  unsigned* h1c = &hist[0][(lane & (QP_HIST_COPIES - 1)) * QP_HIST_STRIDE];
  unsigned* h2c = &hist[1][(lane & (QP_HIST_COPIES - 1)) * QP_HIST_STRIDE];

  // pass 1: high byte
  for (int t = tid; t < n; t += 256) atomicAdd(&h1c[keys[t] >> 8], 1u);
  __syncthreads();
  unsigned v1 = 0;
#pragma unroll
  for (int c = 0; c < QP_HIST_COPIES; ++c) v1 += hist[0][c * QP_HIST_STRIDE + tid];
  const unsigned before1 = block_excl_scan_256(v1, wave_tot[0]);
  if (before1 < (unsigned)k && (unsigned)k <= before1 + v1) { res[0] = (unsigned)tid; res[1] = before1; }
  __syncthreads();
  const unsigned b1 = res[0], c1 = res[1];
  // pass 2: low byte inside bucket b1
  for (int t = tid; t < n; t += 256) { const unsigned key = keys[t]; if ((key >> 8) == b1) atomicAdd(&h2c[key & 255u], 1u); }
  __syncthreads();
  unsigned v2 = 0;
#pragma unroll
  for (int c = 0; c < QP_HIST_COPIES; ++c) v2 += hist[1][c * QP_HIST_STRIDE + tid];
  const unsigned kk = (unsigned)k - c1;
  const unsigned before2 = block_excl_scan_256(v2, wave_tot[1]);
  if (before2 < kk && kk <= before2 + v2) { res[2] = (unsigned)tid; res[3] = before2; }
  __syncthreads();
  const unsigned tau = (b1 << 8) | res[2];
  const unsigned r_ties = (unsigned)k - (c1 + res[3]);   // >= 1 ties (key == tau) to take, lowest index first

  // kept tokens in front of this slice
  unsigned lt = 0, eq = 0;
  for (int t = tid; t < t0; t += 256) { const unsigned key = keys[t]; lt += key < tau; eq += key == tau; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o, 64); eq += __shfl_xor(eq, o, 64); }
  if (lane == 0) { red[0][wave] = lt; red[1][wave] = eq; }
  __syncthreads();
  if (wave == 0) {
    const unsigned lt_b = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const unsigned eq_b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    // slice: lane j < TS owns token t0 + j
    const bool mine = lane < QP_PRUNE_TS && t0 + lane < n;
    const unsigned mykey = mine ? (unsigned)keys[t0 + lane] : 0xffffffffu;
    const bool is_lt = mykey < tau, is_eq = mykey == tau;
    const unsigned long long m_eq = __ballot(is_eq);
    const unsigned eq_rank = eq_b + (unsigned)__popcll(m_eq & ((1ull << lane) - 1ull));
    const bool keep = is_lt || (is_eq && eq_rank < r_ties);
    const unsigned long long m_keep = __ballot(keep);
    const unsigned rank = (unsigned)__popcll(m_keep & ((1ull << lane) - 1ull));
    const unsigned base = lt_b + min(eq_b, r_ties);
    if (keep) { s_tok[rank] = t0 + lane; s_pos[rank] = (int)(base + rank); kept[base + rank] = t0 + lane; }
    if (lane == 0) s_nk = (int)__popcll(m_keep);
  }
  __syncthreads();
  // gather: 16-lane group g moves the K and V rows of kept slot g (16 B per lane), 4 rows in flight
  const int c = tid & 15, grp = tid >> 4;
  if (grp < s_nk) {
    const uint4* ks = k_src + (int64_t)s_tok[grp] * 16 + c;
    const uint4* vs = v_src + (int64_t)s_tok[grp] * 16 + c;
    uint4* kd = k_dst + (dst_row0 + s_pos[grp]) * 16 + c;
    uint4* vd = v_dst + (dst_row0 + s_pos[grp]) * 16 + c;
    for (int h = 0; h < hkv; h += 2) {
      const bool two = h + 1 < hkv;
      const uint4 a0 = ks[h * src_hs16], b0 = vs[h * src_hs16];
      uint4 a1 = a0, b1v = b0;
      if (two) { a1 = ks[(h + 1) * src_hs16]; b1v = vs[(h + 1) * src_hs16]; }
      kd[h * dst_hs16] = a0; vd[h * dst_hs16] = b0;
      if (two) { kd[(h + 1) * dst_hs16] = a1; vd[(h + 1) * dst_hs16] = b1v; }
    }
  }
}

int qp_launch_prune_keys(const uint16_t* norm_keys, int64_t n, int64_t k, const void* k_src, const void* v_src, int64_t src_head_stride,
                         int hkv, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0, int32_t* kept, hipStream_t s) {
  const unsigned grid = (unsigned)((n + QP_PRUNE_TS - 1) / QP_PRUNE_TS);
  prune_keys_kernel<<<grid, 256, 0, s>>>(norm_keys, (int)n, (int)k, (const uint4*)k_src, (const uint4*)v_src, src_head_stride / 8, hkv,
                                         (uint4*)k_dst, (uint4*)v_dst, dst_head_stride / 8, dst_row0, kept);
  return qp_check_launch("prune_keys");
}

// ------------------------------------------------------------------------------------------------
// qp_prune_tail, round 3: the in-place seam (utils.py:266-342 on the arena itself) in TWO launches, no HBM bounce.
//   tail_keys_kernel          one 16-lane group per tail token: the hkv key (or value) rows of the token -> canonical per-head sums
//                             (same order as key_sumsq_kernel) -> heads added in ascending order -> 16-bit norm key; also clears
//                             the hand-shake flags of the second launch.
//   prune_tail_inplace_kernel prune_keys_kernel's select, then the compaction staged through registers: rows [past, past+n) shrink
//                             to [past, past+k) IN PLACE.  Kept token t moves to position pos(t) <= t, so the <= 16 destination
//                             rows of a 16-token slice are source rows of at most two slices, both at or before its own.  Every
//                             workgroup (1) loads the K/V rows of its kept tokens into registers (<= 2*hkv*16 B per lane),
//                             (2) publishes flag[slice] = 1 ("my loads have retired"), (3) waits for the flags of the (<= 2)
//                             LOWER slices whose rows it is about to overwrite, (4) stores.
//   Why the wait cannot deadlock: a slice waits only for lower slices and never the other way round (no cycle), and a not yet
//   dispatched lower slice always finds a slot, because (a) the launcher only uses this kernel when the WHOLE grid fits at once on
//   the CUs THE STREAM MAY USE (its CU mask, hipExtStreamGetCUMask; x workgroups per CU from the occupancy API: n <= 8192 needs
//   <= 512 workgroups, an unmasked MI355X holds 1024), and (b) the context keeps at most ONE such grid in flight — a call on another
//   stream is ordered behind the previous call's event (qp_api.hip: qp_prune_tail) — so the spinning workgroups of other in-place
//   grids can never fill the slots; workgroups of every OTHER kernel finish on their own.  Where (a) fails the staged form runs.
//   (An atomic ticket — slice = start order — would need neither condition, but 360 same-address device atomics serialise at
//   ~25 ns each: +3-4 us per launch, measured in profiles/r3_prune_tail_old_vs_new.json, on a 10 us kernel.)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tail_keys_kernel(const uint4* __restrict__ rows, int64_t hs16, int64_t row0, int n, int hkv,
                                                        uint16_t* __restrict__ norm_keys, int largest, int* __restrict__ sync_words,
                                                        int n_sync_words) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_sync_words; i += gridDim.x * 256) sync_words[i] = 0;
  const int c = threadIdx.x & 15;
  const int t = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (t >= n) return;                                   // whole 16-lane groups leave together
  const uint4* p = rows + (row0 + t) * 16 + c;
  uint4 v[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) if (h < hkv) v[h] = p[h * hs16];
  float tot = 0.f;
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    if (h < hkv) {
      const float s = row16_butterfly(chunk_sumsq(v[h]));
      tot = h == 0 ? s : tot + s;
    }
  }
  if (c == 0) {
    const uint16_t b = f32_to_bf16_bits(sqrt_rn_f32(tot));
    norm_keys[t] = largest ? (uint16_t)~b : b;
  }
}

int qp_launch_tail_keys(const void* rows, int64_t head_stride, int64_t row0, int64_t n, int hkv, uint16_t* norm_keys, int largest,
                        int* sync_words, int n_sync_words, hipStream_t s) {
  tail_keys_kernel<<<(unsigned)((n + 15) / 16), 256, 0, s>>>((const uint4*)rows, head_stride / 8, row0, (int)n, hkv, norm_keys, largest,
                                                            sync_words, n_sync_words);
  return qp_check_launch("tail_keys");
}

// kThreads / 16 tokens per workgroup (one 16-lane group per token)
template <int kThreads>
__global__ __launch_bounds__(kThreads) void prune_tail_inplace_kernel(const uint16_t* __restrict__ keys_g, int n, int k, uint4* k_cache,
                                                                      uint4* v_cache, int64_t hs16, int64_t past, int hkv,
                                                                      int32_t* __restrict__ kept, int* sync_words) {
  constexpr int QP_TAIL_TS = kThreads / 16;
  constexpr int kWaves = kThreads / 64;
  __shared__ __attribute__((aligned(16))) uint16_t keys[QP_PRUNE_MAX_N];
  __shared__ unsigned hist[2][QP_HIST_COPIES * QP_HIST_STRIDE];
  __shared__ unsigned res[8];                  // pass 1: bucket, count before it; pass 2 (res + 4): low byte, count before tau inside the bucket
  __shared__ unsigned red[2][kWaves];
  __shared__ int s_tok[QP_TAIL_TS], s_pos[QP_TAIL_TS];
  __shared__ int s_nk, s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* flags = sync_words;
  const int n8 = n >> 3;
  for (int i = tid; i < n8; i += kThreads) ((uint4*)keys)[i] = ((const uint4*)keys_g)[i];
  if (tid < (n & 7)) keys[n8 * 8 + tid] = keys_g[n8 * 8 + tid];
  for (int i = tid; i < 2 * QP_HIST_COPIES * QP_HIST_STRIDE; i += kThreads) (&hist[0][0])[i] = 0u;
  __syncthreads();
  unsigned* h1c = &hist[0][(lane & (QP_HIST_COPIES - 1)) * QP_HIST_STRIDE];
  unsigned* h2c = &hist[1][(lane & (QP_HIST_COPIES - 1)) * QP_HIST_STRIDE];

  // two-pass 256-bin radix select of the threshold key tau (same tie rule as select_kernel / prune_keys_kernel)
  for (int t = tid; t < n; t += kThreads) atomicAdd(&h1c[keys[t] >> 8], 1u);
  __syncthreads();
  find_bucket_256(&hist[0][0], QP_HIST_COPIES, QP_HIST_STRIDE, (unsigned)k, res);
  const unsigned b1 = res[0], c1 = res[1];
  for (int t = tid; t < n; t += kThreads) { const unsigned key = keys[t]; if ((key >> 8) == b1) atomicAdd(&h2c[key & 255u], 1u); }
  __syncthreads();
  find_bucket_256(&hist[1][0], QP_HIST_COPIES, QP_HIST_STRIDE, (unsigned)k - c1, res + 4);      // (ends with a barrier)
  const unsigned tau = (b1 << 8) | res[4];
  const unsigned r_ties = (unsigned)k - (c1 + res[5]);   // >= 1 ties (key == tau) to take, lowest index first
  const int slice = (int)blockIdx.x;
  const int t0 = slice * QP_TAIL_TS;

  // kept tokens in front of this slice
  unsigned lt = 0, eq = 0;
  for (int t = tid; t < t0; t += kThreads) { const unsigned key = keys[t]; lt += key < tau; eq += key == tau; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o, 64); eq += __shfl_xor(eq, o, 64); }
  if (lane == 0) { red[0][wave] = lt; red[1][wave] = eq; }
  __syncthreads();
  if (wave == 0) {
    unsigned lt_b = 0, eq_b = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { lt_b += red[0][w]; eq_b += red[1][w]; }
    const bool mine = lane < QP_TAIL_TS && t0 + lane < n;                     // lane j owns token t0 + j
    const unsigned mykey = mine ? (unsigned)keys[t0 + lane] : 0xffffffffu;
    const bool is_lt = mykey < tau, is_eq = mykey == tau;
    const unsigned long long m_eq = __ballot(is_eq);
    const unsigned eq_rank = eq_b + (unsigned)__popcll(m_eq & ((1ull << lane) - 1ull));
    const bool keep = is_lt || (is_eq && eq_rank < r_ties);
    const unsigned long long m_keep = __ballot(keep);
    const unsigned rank = (unsigned)__popcll(m_keep & ((1ull << lane) - 1ull));
    const unsigned base = lt_b + min(eq_b, r_ties);
    if (keep) { s_tok[rank] = t0 + lane; s_pos[rank] = (int)(base + rank); kept[base + rank] = t0 + lane; }
    if (lane == 0) { s_nk = (int)__popcll(m_keep); s_base = (int)base; }
  }
  __syncthreads();
  // (2) stage: 16-lane group g holds the K and V rows (all heads) of kept slot g in registers
  const int c = tid & 15, grp = tid >> 4;
  const int nk = s_nk, base = s_base;
  const bool have = grp < nk;
  uint4 rk[8], rv[8];
  if (have) {
    const uint4* ks = k_cache + (past + s_tok[grp]) * 16 + c;
    const uint4* vs = v_cache + (past + s_tok[grp]) * 16 + c;
#pragma unroll
    for (int h = 0; h < 8; ++h) if (h < hkv) { rk[h] = ks[h * hs16]; rv[h] = vs[h * hs16]; }
  }
  // The rows must be IN the registers before anyone may overwrite them.  Each loaded vector is made an operand of an (empty) volatile
  // asm: the data dependence makes the compiler wait for the loads right here, and the volatile flag store below cannot move above it.
  // (A `"memory"` clobber would do the same but forces rk / rv into scratch memory: every row then went to HBM-backed scratch and
  // back — rocprofv3 FETCH_SIZE / WRITE_SIZE read exactly 2x the algorithmic bytes.)
  if (have) {
#pragma unroll
    for (int h = 0; h < 8; ++h)
      if (h < hkv) {
        asm volatile("" : "+v"(rk[h].x), "+v"(rk[h].y), "+v"(rk[h].z), "+v"(rk[h].w));
        asm volatile("" : "+v"(rv[h].x), "+v"(rv[h].y), "+v"(rv[h].z), "+v"(rv[h].w));
      }
  }
  __syncthreads();
  // (3) publish, (4) wait for the lower slices whose source rows [base, base+nk) covers.  RELAXED device-scope atomics: the flag
  // carries no data (it says "my loads have retired", which the s_waitcnt above established locally), so neither side needs the L2
  // write-back / invalidate of a release / acquire pair (measured: 35 us per call with them at n = 5760).
  if (tid == 0) __hip_atomic_store(&flags[slice], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nk > 0 && tid < 2) {
    const int dep = tid == 0 ? base / QP_TAIL_TS : (base + nk - 1) / QP_TAIL_TS;
    if (dep < slice)
      while (__hip_atomic_load(&flags[dep], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
  }
  __syncthreads();
  // (5) store
  if (have) {
    uint4* kd = k_cache + (past + s_pos[grp]) * 16 + c;
    uint4* vd = v_cache + (past + s_pos[grp]) * 16 + c;
#pragma unroll
    for (int h = 0; h < 8; ++h) if (h < hkv) { kd[h * hs16] = rk[h]; vd[h * hs16] = rv[h]; }
  }
}

// workgroups of prune_tail_inplace_kernel<256> the device holds at once (0 = unknown: the caller uses the staged form)
int qp_prune_tail_inplace_capacity(int cus) {
  static std::atomic<int> per_cu{-1};
  int v = per_cu.load(std::memory_order_relaxed);
  if (v < 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)prune_tail_inplace_kernel<256>, 256, 0) != hipSuccess) nb = 0;
    per_cu.store(v = nb, std::memory_order_relaxed);
  }
  return v * cus;
}

int qp_launch_prune_tail_inplace(const uint16_t* norm_keys, int64_t n, int64_t k, void* k_cache, void* v_cache, int64_t head_stride,
                                 int64_t past_len, int hkv, int32_t* kept, int* sync_words, hipStream_t s) {
  prune_tail_inplace_kernel<256><<<(unsigned)((n + 15) / 16), 256, 0, s>>>(norm_keys, (int)n, (int)k, (uint4*)k_cache, (uint4*)v_cache,
                                                                         head_stride / 8, past_len, hkv, kept, sync_words);
  return qp_check_launch("prune_tail_inplace");
}

static size_t select_smem_bytes(int64_t n) { return (256 + 256 + 16 + 4 + 4) * 4 + (size_t)((n + 7) / 8 * 8) * 2; }

int qp_launch_select(const float* head_sumsq, int n_heads, int64_t n, int64_t k, int32_t* kept, uint16_t* norm_bits,
                     void* ws, int largest, hipStream_t s, const uint16_t* keys_in) {
  if (n > 65536) {
    if (ws == nullptr && keys_in == nullptr)
      return qp_fail(QP_ERR_WORKSPACE, "select: n=%lld > 65536 needs a workspace of qp_select_workspace_bytes(n)", (long long)n);
    select_kernel<false><<<1, 1024, select_smem_bytes(0), s>>>(head_sumsq, n_heads, (int)n, (int)k, kept, norm_bits, (uint16_t*)ws, largest, keys_in);
    return qp_check_launch("select(global keys)");
  }
  size_t smem = select_smem_bytes(n);
  static std::atomic<unsigned long long> lds_ok{0};
  if (int rc = qp_opt_in_lds(lds_ok, (const void*)select_kernel<true>, 160 * 1024 - 64, "select")) return rc;
  select_kernel<true><<<1, 1024, smem, s>>>(head_sumsq, n_heads, (int)n, (int)k, kept, norm_bits, nullptr, largest, keys_in);
  return qp_check_launch("select");
}

// ------------------------------------------------------------------------------------------------
// K6: gather kept rows (staging or arena tail -> arena), K and V in one launch.
//   256-B rows, 16 lanes x 16 B per row, 4 independent rows in flight per thread.
// ------------------------------------------------------------------------------------------------
template <bool kIndexed>
__global__ __launch_bounds__(256) void gather_kv_kernel(const uint4* __restrict__ k_src, const uint4* __restrict__ v_src,
                                                        int64_t src_hs16, const int32_t* __restrict__ idx, int64_t k,
                                                        int hkv, uint4* __restrict__ k_dst, uint4* __restrict__ v_dst,
                                                        int64_t dst_hs16, int64_t dst_row0) {
  const int c = threadIdx.x & 15;
  const int64_t rows = 2 * (int64_t)hkv * k;          // (kv, head, j)
  const int64_t stride = (int64_t)gridDim.x * 16;
  int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  for (; r < rows; r += 4 * stride) {
    uint4 v[4];
    int64_t dsto[4];
    bool ok[4], isv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int64_t rr = r + u * stride;
      ok[u] = rr < rows;
      if (!ok[u]) rr = rows - 1;
      int64_t j = rr % k, hh = (rr / k) % hkv;
      isv[u] = rr >= (int64_t)hkv * k;
      int64_t srow = kIndexed ? (int64_t)idx[j] : j;
      const uint4* src = isv[u] ? v_src : k_src;
      v[u] = src[hh * src_hs16 + srow * 16 + c];
      dsto[u] = hh * dst_hs16 + (dst_row0 + j) * 16 + c;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) (isv[u] ? v_dst : k_dst)[dsto[u]] = v[u];
  }
}

static int gather_grid(int64_t rows) {
  int64_t blocks = (rows + 63) / 64;   // 16 rows per block pass x 4 unrolled
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

int qp_launch_gather_kv(const void* k_src, const void* v_src, int64_t src_head_stride, const int32_t* idx, int64_t k,
                        int hkv, void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0, hipStream_t s) {
  gather_kv_kernel<true><<<gather_grid(2 * hkv * k), 256, 0, s>>>((const uint4*)k_src, (const uint4*)v_src, src_head_stride / 8,
                                                                   idx, k, hkv, (uint4*)k_dst, (uint4*)v_dst,
                                                                   dst_head_stride / 8, dst_row0);
  return qp_check_launch("gather_kv");
}

int qp_launch_copy_rows_kv(const void* k_src, const void* v_src, int64_t src_head_stride, int64_t k, int hkv, void* k_dst,
                           void* v_dst, int64_t dst_head_stride, int64_t dst_row0, hipStream_t s) {
  gather_kv_kernel<false><<<gather_grid(2 * hkv * k), 256, 0, s>>>((const uint4*)k_src, (const uint4*)v_src, src_head_stride / 8,
                                                                    nullptr, k, hkv, (uint4*)k_dst, (uint4*)v_dst,
                                                                    dst_head_stride / 8, dst_row0);
  return qp_check_launch("copy_rows_kv");
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const int32_t* __restrict__ idx, int64_t k,
                                                          int64_t row16, uint4* __restrict__ dst) {
  const int64_t total = k * row16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t j = i / row16, c = i - j * row16;
    dst[i] = src[(int64_t)idx[j] * row16 + c];
  }
}

int qp_launch_gather_rows(const void* src, const int32_t* idx, int64_t k, int64_t row_bytes, void* dst, hipStream_t s) {
  int64_t total = k * (row_bytes / 16);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  gather_rows_kernel<<<(int)blocks, 256, 0, s>>>((const uint4*)src, idx, k, row_bytes / 16, (uint4*)dst);
  return qp_check_launch("gather_rows");
}

// ------------------------------------------------------------------------------------------------
// Group-token parallel receive side: all-gathered [rank][K | V | sums] blocks -> staging block in token order.
// One thread per 16 B of a K/V row (256-B rows: 16 threads per row); the last grid rows carry the key sums.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_unpack_kernel(const unsigned char* __restrict__ g, int world, int hkv, int m2, int64_t n,
                                                        uint4* __restrict__ ks, uint4* __restrict__ vs, int64_t stage_hs16,
                                                        float* __restrict__ ss) {
  const int64_t m = 2 * (int64_t)m2;
  const int64_t kv_bytes = (int64_t)hkv * m * 256, chunk = 2 * kv_bytes + (int64_t)hkv * m * 4;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);          // (kv, head, token)
  const int slot = threadIdx.x & 15;
  const int64_t rows_total = 2 * (int64_t)hkv * n;
  if (row < rows_total) {
    const int kv = (int)(row / (hkv * n));
    const int64_t rem = row - (int64_t)kv * hkv * n;
    const int h = (int)(rem / n);
    const int64_t t = rem - (int64_t)h * n;
    const int c = (int)(t / m2);                                              // zigzag chunk of the token
    const int r = c < world ? c : 2 * world - 1 - c, half = c < world ? 0 : 1;
    const int64_t srow = (int64_t)h * m + half * m2 + (t - (int64_t)c * m2);
    const uint4 v = *reinterpret_cast<const uint4*>(g + r * chunk + kv * kv_bytes + srow * 256 + slot * 16);
    (kv ? vs : ks)[(int64_t)h * stage_hs16 + t * 16 + slot] = v;
  }
  // key sums: threads of the whole grid stride over [hkv][n]
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)hkv * n; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i / n);
    const int64_t t = i - (int64_t)h * n;
    const int c = (int)(t / m2);
    const int r = c < world ? c : 2 * world - 1 - c, half = c < world ? 0 : 1;
    ss[i] = *reinterpret_cast<const float*>(g + r * chunk + 2 * kv_bytes + ((int64_t)h * m + half * m2 + (t - (int64_t)c * m2)) * 4);
  }
}

int qp_launch_sp_unpack(const void* gathered, int world, int hkv, int64_t m2, int64_t n, void* k_stage, void* v_stage,
                        int64_t stage_head_stride, float* sumsq_out, hipStream_t s) {
  const int64_t rows = 2 * (int64_t)hkv * n;
  sp_unpack_kernel<<<(unsigned)((rows + 15) / 16), 256, 0, s>>>((const unsigned char*)gathered, world, hkv, (int)m2, n, (uint4*)k_stage,
                                                              (uint4*)v_stage, stage_head_stride / 8, sumsq_out);
  return qp_check_launch("sp_unpack");
}
