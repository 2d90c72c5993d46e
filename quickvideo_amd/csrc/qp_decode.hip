// Decode step (a10: generate over the pruned cache, qwen25_lvu.py:744-761): every kernel of one greedy decode step takes its
// scalars (cache length, rotary position) from a DEVICE state block, so the step has no host-side arguments that change
// between tokens and the whole step is captured once in a hipGraph and replayed per token.
//   state int64[2] = { kv_len  : rows already stored in every layer's K/V cache,
//                      rope_pos: position id of the token being decoded (same on the three M-RoPE streams: text token) }
// Kernels:
//   gemv_kernel          out[N] = epilogue(W[N][K] . x[K]); the weight stream is the roofline of the step (HBM, 2 B/weight).
//                        Optional prologue: x = RMSNorm(h) * w (same arithmetic and summation order as add_rmsnorm_kernel).
//                        Epilogues: + bias | SwiGLU of a (gate,up) row pair | + residual (in place) - one bf16 rounding per
//                        torch op of the reference, like the unfused prefill kernels.
//   decode_rope_kernel   M-RoPE of the single token (cos/sin computed inline exactly like mrope_table_kernel), K/V appended at
//                        row kv_len.
//   decode_attn_kernel   single-query GQA attention over rows [0, kv_len]: fixed grid (splits x kv heads), every K/V row read
//                        once for the whole q-head group; partial (m, l, o) per split, merged by decode_combine_kernel.
//   decode_advance_kernel  state += 1 (the per-layer state blocks of a model are one array).
#include <cstdlib>
#include "qp_attn.h"

namespace {

using qpattn::lds_read_b128;
using qpattn::lds_read_tr16;
using qpattn::s16x8_t;
using qpattn::xhalf_max;
using qpattn::xhalf_sum;

__device__ __forceinline__ float dot8(uint4 w, uint4 x, float acc) {
  const unsigned ww[4] = {w.x, w.y, w.z, w.w}, xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc = __builtin_fmaf(__uint_as_float(ww[i] << 16), __uint_as_float(xw[i] << 16), acc);
    acc = __builtin_fmaf(__uint_as_float(ww[i] & 0xffff0000u), __uint_as_float(xw[i] & 0xffff0000u), acc);
  }
  return acc;
}

typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-byte load: weights are read once per token and must not evict the K/V rows from L2
__device__ __forceinline__ uint4 load_stream(const uint4* p) {
  const u32x4n v = __builtin_nontemporal_load((const u32x4n*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

enum { GEMV_BIAS = 0, GEMV_SWIGLU = 1, GEMV_RESIDUAL = 2 };

// W bf16 [N (x2 for SwiGLU: gate rows then up rows)][K]; x (or h when norm_w != NULL) bf16 [K]; out bf16 [N].
// One wave per work item, whole K per wave (no cross-wave sums: K-split variants with LDS partial sums measured 7-35 % slower
// per token).  PAIR: an item is two rows sharing every x chunk read from LDS (SwiGLU: gate row r and up row r + N; otherwise
// rows 2p, 2p + 1); !PAIR: one row (small N: twice the work items).  Loads run one batch (4 x 16 B per row and lane) ahead of
// the arithmetic in two alternating register sets, and the first batch is issued BEFORE the x / RMSNorm prologue (the weights
// do not depend on x), so the prologue's latency hides under it.
template <int MODE, bool PAIR>
__global__ __launch_bounds__(256) void gemv_kernel(const uint4* __restrict__ W, const uint4* __restrict__ xin,
                                                   const uint4* __restrict__ norm_w, const uint16_t* __restrict__ bias,
                                                   uint16_t* out, int N, int K16, int iters, float inv_hidden, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* xs = (uint4*)smem;
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_items = MODE == GEMV_SWIGLU ? N : (PAIR ? (N + 1) / 2 : N);

  uint4 wA[4], wA2[4], wB[4], wB2[4];            // two register sets: one consumed while the other is in flight
  auto rows_of = [&](int j, int& ra, int& rb, bool& live_a, bool& live_b) {
    const int it = j * 4 + wave;
    live_a = it < n_items;
    if (MODE == GEMV_SWIGLU) { ra = it; rb = it + N; live_b = live_a; }
    else if (PAIR) { ra = 2 * it; rb = 2 * it + 1; live_b = live_a && rb < N; }
    else { ra = it; rb = 0; live_b = false; }
  };
  // branch-free batch of loads: chunks past the end of the row re-read its last chunk (their x is zeroed instead), rows past
  // the end re-read row 0 (their result is dropped)
  auto issue = [&](int j, int cb, uint4 (&w)[4], uint4 (&w2)[4]) {
    int ra, rb; bool la, lb;
    rows_of(j, ra, rb, la, lb);
    const uint4* wa = W + (int64_t)(la ? ra : 0) * K16;
    const uint4* wb = W + (int64_t)(lb ? rb : 0) * K16;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int cc = cb + lane + 64 * u;
      cc = cc < K16 ? cc : K16 - 1;
      w[u] = load_stream(wa + cc);
      if (PAIR) w2[u] = load_stream(wb + cc);
    }
  };
  int j = blockIdx.x, cb = 0;
  bool have = j < iters;
  if (have) issue(j, cb, wA, wA2);               // before the prologue

  if (norm_w) {                                  // x = RMSNorm(h) * w, arithmetic of add_rmsnorm_kernel (delta == NULL)
    float ss = 0.f;
    for (int c = tid; c < K16; c += 256) {
      const uint4 x = xin[c];
      xs[c] = x;
      const unsigned xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = __uint_as_float(xw[i] << 16), hi = __uint_as_float(xw[i] & 0xffff0000u);
        ss = __builtin_fmaf(lo, lo, ss);
        ss = __builtin_fmaf(hi, hi, ss);
      }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    const float rs = 1.0f / __fsqrt_rn(tot * inv_hidden + eps);
    for (int c = tid; c < K16; c += 256) {
      const uint4 x = xs[c], ww = norm_w[c];
      const unsigned xw[4] = {x.x, x.y, x.z, x.w}, wv[4] = {ww.x, ww.y, ww.z, ww.w};
      unsigned o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = round_bf16(__uint_as_float(xw[i] << 16) * rs) * __uint_as_float(wv[i] << 16);
        const float hi = round_bf16(__uint_as_float(xw[i] & 0xffff0000u) * rs) * __uint_as_float(wv[i] & 0xffff0000u);
        o[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
      }
      xs[c] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  } else {
    for (int c = tid; c < K16; c += 256) xs[c] = xin[c];
  }
  __syncthreads();

  float acc = 0.f, acc2 = 0.f;
  // consume the batch in (cw, cw2) while the next one is loaded into (nw, nw2)
  auto step = [&](uint4 (&cw)[4], uint4 (&cw2)[4], uint4 (&nw)[4], uint4 (&nw2)[4]) {
    const int jc = j, cbc = cb;
    cb += 256;
    const bool row_done = cb >= K16;
    if (row_done) { cb = 0; j += gridDim.x; }
    have = j < iters;
    if (have) issue(j, cb, nw, nw2);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cc = cbc + lane + 64 * u;
      uint4 x = xs[cc < K16 ? cc : K16 - 1];
      if (cc >= K16) x = make_uint4(0, 0, 0, 0);
      acc = dot8(cw[u], x, acc);
      if (PAIR) acc2 = dot8(cw2[u], x, acc2);
    }
    if (row_done) {
      int ra, rb; bool la, lb;
      rows_of(jc, ra, rb, la, lb);
      acc = wave_sum(acc);
      if (PAIR) acc2 = wave_sum(acc2);
      if (lane == 0) {
        if (MODE == GEMV_SWIGLU) {
          if (la) {
            const float g = round_bf16(acc), u = round_bf16(acc2);
            const float a = round_bf16(g / (1.0f + __expf(-g)));          // silu_mul_bf16 of qp_elementwise.hip
            out[ra] = f32_to_bf16_bits(a * u);
          }
        } else if (MODE == GEMV_BIAS) {
          if (la) out[ra] = f32_to_bf16_bits(bias ? acc + bf16_bits_to_f32(bias[ra]) : acc);
          if (lb) out[rb] = f32_to_bf16_bits(bias ? acc2 + bf16_bits_to_f32(bias[rb]) : acc2);
        } else {
          if (la) out[ra] = f32_to_bf16_bits(bf16_bits_to_f32(out[ra]) + round_bf16(acc));
          if (lb) out[rb] = f32_to_bf16_bits(bf16_bits_to_f32(out[rb]) + round_bf16(acc2));
        }
      }
      acc = 0.f; acc2 = 0.f;
    }
  };
  while (have) {
    step(wA, wA2, wB, wB2);
    if (!have) break;
    step(wB, wB2, wA, wA2);
  }
}

// cos_t / sin_t: bf16 [64] table of the token's position (qp_mrope_table, computed once per token and shared by all layers);
// NULL -> computed here from state[1] with the same arithmetic.
__global__ __launch_bounds__(256) void decode_rope_kernel(const uint4* __restrict__ qkv, const int64_t* __restrict__ state,
                                                          const uint4* __restrict__ cos_t, const uint4* __restrict__ sin_t,
                                                          float theta, int hq, int hkv, uint4* __restrict__ q_out,
                                                          uint4* __restrict__ k_cache, uint4* __restrict__ v_cache,
                                                          int64_t hs16) {
  const int c = threadIdx.x & 15;
  const int hr = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (hr >= hq + 2 * hkv) return;
  const int64_t row = state[0];
  const uint4 x = qkv[hr * 16 + c];
  if (hr >= hq + hkv) {
    v_cache[(int64_t)(hr - hq - hkv) * hs16 + row * 16 + c] = x;
    return;
  }
  const float posf = cos_t ? 0.f : (float)state[1];
  uint4 ct = make_uint4(0, 0, 0, 0), st = ct;
  if (cos_t) { ct = cos_t[c & 7]; st = sin_t[c & 7]; }
  const unsigned cw[4] = {ct.x, ct.y, ct.z, ct.w}, sw[4] = {st.x, st.y, st.z, st.w};
  uint4 p;
  p.x = __shfl_xor((int)x.x, 8, 16); p.y = __shfl_xor((int)x.y, 8, 16);
  p.z = __shfl_xor((int)x.z, 8, 16); p.w = __shfl_xor((int)x.w, 8, 16);
  const float sign = (c < 8) ? -1.f : 1.f;
  const unsigned xw[4] = {x.x, x.y, x.z, x.w}, pw[4] = {p.x, p.y, p.z, p.w};
  unsigned short o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int w = e >> 1, sh = (e & 1) * 16;
    float cv, sv;
    if (cos_t) {
      cv = bf16_bits_to_f32((unsigned short)(cw[w] >> sh)); sv = bf16_bits_to_f32((unsigned short)(sw[w] >> sh));
    } else {
      const int f = (c & 7) * 8 + e;                                        // frequency index < 64 (mrope_table_kernel)
      const float inv_freq = 1.0f / powf(theta, (float)(2 * f) / 128.0f);
      const float ang = posf * inv_freq;
      cv = round_bf16(cosf(ang)); sv = round_bf16(sinf(ang));
    }
    const float xv = bf16_bits_to_f32((unsigned short)(xw[w] >> sh)), pv = bf16_bits_to_f32((unsigned short)(pw[w] >> sh));
    const float a = round_bf16(xv * cv);
    const float b = round_bf16((sign * pv) * sv);
    o[e] = f32_to_bf16_bits(a + b);
  }
  uint4 ov;
  ov.x = o[0] | ((unsigned)o[1] << 16); ov.y = o[2] | ((unsigned)o[3] << 16);
  ov.z = o[4] | ((unsigned)o[5] << 16); ov.w = o[6] | ((unsigned)o[7] << 16);
  if (hr < hq) q_out[hr * 16 + c] = ov;
  else k_cache[(int64_t)(hr - hq) * hs16 + row * 16 + c] = ov;
}

constexpr int kDecPartial = 130;          // floats per (q head, split): m, l, o[128]

// sum over the 16 lanes of a DPP row (all lanes end with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

// grid (nsplit, hkv), 256 threads: 16 key slots (4 waves x 4 DPP rows), lane c of a row owns dims 8c..8c+7 of K and V.
template <int G>
__global__ __launch_bounds__(256) void decode_attn_kernel(const uint4* __restrict__ q, const uint4* __restrict__ k_cache,
                                                          const uint4* __restrict__ v_cache, int64_t hs16,
                                                          const int64_t* __restrict__ state, float c_log2,
                                                          float* __restrict__ ws) {
  __shared__ float lds_m[32], lds_l[128];
  __shared__ __attribute__((aligned(16))) float lds_o[16 * G * 128];
  const int tid = threadIdx.x, c = tid & 15, slot = tid >> 4;
  const int split = blockIdx.x, nsplit = gridDim.x, kvh = blockIdx.y;
  const int64_t L = state[0] + 1;                              // the token's own K/V row is already appended
  int64_t chunk = (L + nsplit - 1) / nsplit;
  chunk = (chunk + 15) & ~(int64_t)15;
  const int64_t k0 = (int64_t)split * chunk;
  int64_t k1 = k0 + chunk;
  if (k1 > L) k1 = L;
  float qf[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    const uint4 x = q[(kvh * G + h) * 16 + c];
    const unsigned xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { qf[h][2 * i] = __uint_as_float(xw[i] << 16); qf[h][2 * i + 1] = __uint_as_float(xw[i] & 0xffff0000u); }
  }
  float m[G], l[G], acc[G][8];
#pragma unroll
  for (int h = 0; h < G; ++h) {
    m[h] = -1e30f; l[h] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[h][e] = 0.f;
  }
  const uint4* kb = k_cache + (int64_t)kvh * hs16;
  const uint4* vb = v_cache + (int64_t)kvh * hs16;
  for (int64_t base = k0; base < k1; base += 64) {             // a round = 64 keys: 4 per slot, all 8 loads issued up front
    uint4 kx[4], vx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t key = base + 16 * u + slot;
      kx[u] = vx[u] = make_uint4(0, 0, 0, 0);
      if (key < k1) { kx[u] = kb[key * 16 + c]; vx[u] = vb[key * 16 + c]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (base + 16 * u >= k1) break;                          // workgroup-uniform
      const bool live = base + 16 * u + slot < k1;
      const unsigned kw[4] = {kx[u].x, kx[u].y, kx[u].z, kx[u].w}, vw[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
      float kf[8], vf[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kf[2 * i] = __uint_as_float(kw[i] << 16); kf[2 * i + 1] = __uint_as_float(kw[i] & 0xffff0000u);
        vf[2 * i] = __uint_as_float(vw[i] << 16); vf[2 * i + 1] = __uint_as_float(vw[i] & 0xffff0000u);
      }
#pragma unroll
      for (int h = 0; h < G; ++h) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = __builtin_fmaf(qf[h][e], kf[e], s);
        s = row16_sum(s);
        s = live ? s : -INFINITY;
        if (__any(s > m[h])) {                                 // lazy reference switch (wave-uniform branch)
          const float mn = fmaxf(m[h], s);
          const float alpha = __builtin_amdgcn_exp2f((m[h] - mn) * c_log2);
          m[h] = mn; l[h] *= alpha;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[h][e] *= alpha;
        }
        const float p = __builtin_amdgcn_exp2f((s - m[h]) * c_log2);
        l[h] += p;
        const float pb = round_bf16(p);                        // P rounded to bf16 before P.V, like the prefill kernel
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[h][e] = __builtin_fmaf(pb, vf[e], acc[h][e]);
      }
    }
  }
  // merge the 16 key slots: one reference per head for the whole workgroup, then a plain sum of the rescaled slot results
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int h = 0; h < G; ++h) {
    float mw = fmaxf(m[h], __shfl_xor(m[h], 16, 64));
    mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
    if (lane == 0) lds_m[wave * 8 + h] = mw;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < G; ++h) {
    const float M = fmaxf(fmaxf(lds_m[h], lds_m[8 + h]), fmaxf(lds_m[16 + h], lds_m[24 + h]));
    const float a = __builtin_amdgcn_exp2f((m[h] - M) * c_log2);
    if (c == 0) lds_l[slot * 8 + h] = l[h] * a;
    float4* dst4 = (float4*)&lds_o[(slot * G + h) * 128 + c * 8];
    dst4[0] = make_float4(acc[h][0] * a, acc[h][1] * a, acc[h][2] * a, acc[h][3] * a);
    dst4[1] = make_float4(acc[h][4] * a, acc[h][5] * a, acc[h][6] * a, acc[h][7] * a);
    m[h] = M;
  }
  __syncthreads();
  for (int o = tid; o < G * 128; o += 256) {                   // o = h * 128 + d
    float ot = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) ot += lds_o[sl * G * 128 + o];
    const int h = o >> 7, d = o & 127;
    ws[((int64_t)(kvh * G + h) * nsplit + split) * kDecPartial + 2 + d] = ot;
  }
  if (tid < G) {
    float lt = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) lt += lds_l[sl * 8 + tid];
    float* dst = ws + ((int64_t)(kvh * G + tid) * nsplit + split) * kDecPartial;
    dst[0] = fmaxf(fmaxf(lds_m[tid], lds_m[8 + tid]), fmaxf(lds_m[16 + tid], lds_m[24 + tid]));
    dst[1] = lt;
  }
}

// MFMA form of the single-query attention (same partial format and combine kernel as decode_attn_kernel): the G query heads
// of a kv head are the first G of the 32 "query columns" of v_mfma_f32_32x32x16_bf16 (the rest are zero: the matrix pipe has
// 20x the throughput this needs, what matters is that one K/V row costs no VALU work).  Every WAVE walks its own 32-key tiles:
// coalesced 16-B global loads -> the K / V LDS images of the prefill kernels (K row-major with XOR-swizzled 16-B slots, V as
// [key/4][d/32][key%4][32] for ds_read_b64_tr_b16) in a wave-private 16 KB region (no workgroup barrier in the loop) ->
// S^T = K.Q^T, online softmax with the lane owning one head, O^T += V^T.P exactly like attn_fwd_kernel_s4.
//
// kFused: the kernel also does the M-RoPE + KV append of the token (one launch less per layer).  `q` is then the RAW fused
// projection [(hq + 2 hkv)][128]; every workgroup rotates the G query heads of its kv head into an LDS tile (the arithmetic
// of decode_rope_kernel with the per-token cos/sin table), and the workgroup whose key range holds row state[0] rotates and
// stores the new K row and copies the V row into the cache before anyone reads it (workgroup-scope fence + barrier; no
// other workgroup touches that row).
template <bool kFused>
__global__ __launch_bounds__(256, 2) void decode_attn_mfma_kernel(const uint4* __restrict__ q, const uint4* __restrict__ cos_t,
                                                                  const uint4* __restrict__ sin_t, uint4* k_cache, uint4* v_cache,
                                                                  int64_t hs16, const int64_t* __restrict__ state, int hq, int hkv,
                                                                  float c, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 16384];
  __shared__ float lds_m[32], lds_l[32], lds_M[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int split = blockIdx.x, nsplit = gridDim.x, kvh = blockIdx.y, G = hq / hkv;
  const int64_t row_new = state[0];
  const int64_t L = row_new + 1;                               // the token's own K/V row counts (appended here or by the caller)
  int64_t chunk = (L + nsplit - 1) / nsplit;
  chunk = (chunk + 31) & ~(int64_t)31;
  const int64_t k0 = (int64_t)split * chunk;
  int64_t k1 = k0 + chunk;
  if (k1 > L) k1 = L;
  unsigned char* kl = lds + wave * 16384;
  unsigned char* vl = kl + 8192;
  bf16x8_t qf[8];
  if (kFused) {
    // 16 lanes per head row: rows 0..G-1 = the query heads, row 8 = the key head, row 9 = the value head of this kv head
    const int r = tid >> 4, cc = tid & 15;
    const bool is_q = r < G, is_k = r == 8, is_v = r == 9;
    const int src = is_q ? kvh * G + r : (is_k ? hq + kvh : hq + hkv + kvh);
    uint4 x = make_uint4(0, 0, 0, 0);
    if (is_q || is_k || is_v) x = q[src * 16 + cc];
    const uint4 ct = cos_t[cc & 7], st = sin_t[cc & 7];
    uint4 pr;
    pr.x = __shfl_xor((int)x.x, 8, 16); pr.y = __shfl_xor((int)x.y, 8, 16);
    pr.z = __shfl_xor((int)x.z, 8, 16); pr.w = __shfl_xor((int)x.w, 8, 16);
    const float sign = (cc < 8) ? -1.f : 1.f;
    const unsigned xw[4] = {x.x, x.y, x.z, x.w}, pw[4] = {pr.x, pr.y, pr.z, pr.w}, cw[4] = {ct.x, ct.y, ct.z, ct.w},
                   sw[4] = {st.x, st.y, st.z, st.w};
    unsigned short o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int w = e >> 1, sh = (e & 1) * 16;
      const float xv = bf16_bits_to_f32((unsigned short)(xw[w] >> sh)), pv = bf16_bits_to_f32((unsigned short)(pw[w] >> sh));
      const float cv = bf16_bits_to_f32((unsigned short)(cw[w] >> sh)), sv = bf16_bits_to_f32((unsigned short)(sw[w] >> sh));
      o8[e] = f32_to_bf16_bits(round_bf16(xv * cv) + round_bf16((sign * pv) * sv));
    }
    uint4 rot;
    rot.x = o8[0] | ((unsigned)o8[1] << 16); rot.y = o8[2] | ((unsigned)o8[3] << 16);
    rot.z = o8[4] | ((unsigned)o8[5] << 16); rot.w = o8[6] | ((unsigned)o8[7] << 16);
    uint4* qt = reinterpret_cast<uint4*>(lds);                 // query tile [32 rows][16 x 16 B], rows >= G zero
    qt[r * 16 + cc] = is_q ? rot : make_uint4(0, 0, 0, 0);
    qt[(16 + r) * 16 + cc] = make_uint4(0, 0, 0, 0);
    if (row_new >= k0 && row_new < k1) {                       // this workgroup owns the new row (workgroup-uniform)
      if (is_k) k_cache[(int64_t)kvh * hs16 + row_new * 16 + cc] = rot;
      if (is_v) v_cache[(int64_t)kvh * hs16 + row_new * 16 + cc] = x;
      __threadfence_block();
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = lds_read_b128(lds, l31 * 256 + (kk * 2 + hi) * 16);
    __syncthreads();                                           // the tile regions alias the query tile
  } else {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      uint4 x = make_uint4(0, 0, 0, 0);
      if (l31 < G) x = q[(kvh * G + l31) * 16 + kk * 2 + hi];
      qf[kk] = __builtin_bit_cast(bf16x8_t, x);
    }
  }
  const int r4 = lane >> 4, slot16 = lane & 15;
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = l31 * 256 + (((kk * 2 + hi) ^ (l31 & 15)) << 4);
  const int voff = (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3) + (hi << 10);
  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x16_t){0};
  float m_run = -1e30f, l_run = 0.f;
  const uint4* kb = k_cache + (int64_t)kvh * hs16;
  const uint4* vb = v_cache + (int64_t)kvh * hs16;
  for (int64_t t0 = k0 + wave * 32; t0 < k1; t0 += 128) {
    uint4 kx[8], vx[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t key = t0 + r4 + 4 * it;
      kx[it] = vx[it] = make_uint4(0, 0, 0, 0);
      if (key < k1) { kx[it] = kb[key * 16 + slot16]; vx[it] = vb[key * 16 + slot16]; }
    }
    __builtin_amdgcn_wave_barrier();                           // the previous tile's fragment reads stay ahead of these writes
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = r4 + 4 * it;
      *reinterpret_cast<uint4*>(kl + row * 256 + ((slot16 ^ (row & 15)) << 4)) = kx[it];
      *reinterpret_cast<uint4*>(vl + (((row >> 2) * 4 + (slot16 >> 2)) << 8) + ((row & 3) << 6) + ((slot16 & 3) << 4)) = vx[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x16_t s = (f32x16_t){0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lds_read_b128(kl, koff[kk]), qf[kk], s, 0, 0, 0);
    if (t0 + 32 > k1) {                                        // ragged last tile (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t jk = t0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        s[r] = jk < k1 ? s[r] : -INFINITY;
      }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = xhalf_max(mx);
    if (!__all((mx - m_run) * c <= 8.0f)) {                    // deferred reference switch, as in the prefill kernels
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    const float mc = m_run * c;
    float rs = 0.f;
    bf16x8_t pf[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[cc * 8 + e], c, -mc));
        rs += pe;
        pf[cc][e] = (__bf16)pe;
      }
    l_run += xhalf_sum(rs);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int off = voff + (((cc * 4) * 4 + db) << 8);     // 4-key row group kq = cc*4 + hi (+2 for the second half)
        const s16x4_t v0 = lds_read_tr16(vl, off);
        const s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));
        const s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[cc], o[db], 0, 0, 0);
      }
  }
  // merge the four waves: one reference per head, rescaled results summed through LDS (the tile regions are free now)
  if (l31 < G && hi == 0) lds_m[wave * 8 + l31] = m_run;
  __syncthreads();
  float* obuf = reinterpret_cast<float*>(lds);                 // [wave][8 heads][128]
  if (l31 < G) {
    const float M = fmaxf(fmaxf(lds_m[l31], lds_m[8 + l31]), fmaxf(lds_m[16 + l31], lds_m[24 + l31]));
    const float a = __builtin_amdgcn_exp2f((m_run - M) * c);
    if (hi == 0) { lds_l[wave * 8 + l31] = l_run * a; if (wave == 0) lds_M[l31] = M; }
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)                           // d = db*32 + 8*q4 + 4*hi + j
        *reinterpret_cast<float4*>(obuf + (wave * 8 + l31) * 128 + db * 32 + 8 * q4 + 4 * hi) =
            make_float4(o[db][q4 * 4 + 0] * a, o[db][q4 * 4 + 1] * a, o[db][q4 * 4 + 2] * a, o[db][q4 * 4 + 3] * a);
  }
  __syncthreads();
  for (int oi = tid; oi < G * 128; oi += 256) {
    const int h = oi >> 7, d = oi & 127;
    const float ot = (obuf[h * 128 + d] + obuf[(8 + h) * 128 + d]) + (obuf[(16 + h) * 128 + d] + obuf[(24 + h) * 128 + d]);
    ws[((int64_t)(kvh * G + h) * nsplit + split) * kDecPartial + 2 + d] = ot;
  }
  if (tid < G) {
    float* dst = ws + ((int64_t)(kvh * G + tid) * nsplit + split) * kDecPartial;
    dst[0] = lds_M[tid];
    dst[1] = (lds_l[tid] + lds_l[8 + tid]) + (lds_l[16 + tid] + lds_l[24 + tid]);
  }
}

// grid hq, 256 threads: out[h][d] = sum_s o_s[d] 2^((m_s - M) c) / sum_s l_s 2^((m_s - M) c), rounded to bf16.
// The split weights are computed once (thread s), then the two halves of the block sum the even / odd splits of dim d.
__global__ __launch_bounds__(256) void decode_combine_kernel(const float* __restrict__ ws, int nsplit, float c_log2,
                                                             uint16_t* __restrict__ out) {
  __shared__ float sa[256], red[4], half_o[128];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = ws + (int64_t)h * nsplit * kDecPartial;
  const float ms = tid < nsplit ? src[tid * kDecPartial] : -1e30f;
  const float ls = tid < nsplit ? src[tid * kDecPartial + 1] : 0.f;
  float M = ms;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
  if (lane == 0) red[wave] = M;
  __syncthreads();
  M = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float a = __builtin_amdgcn_exp2f((ms - M) * c_log2);
  sa[tid] = a;
  float lt = wave_sum(ls * a);
  __syncthreads();                                             // red[] read by everyone, sa[] complete
  if (lane == 0) red[wave] = lt;
  __syncthreads();
  lt = (red[0] + red[1]) + (red[2] + red[3]);
  const int d = tid & 127, par = tid >> 7;
  float ot = 0.f;
#pragma unroll 8
  for (int s = par; s < nsplit; s += 2) ot = __builtin_fmaf(src[s * kDecPartial + 2 + d], sa[s], ot);
  if (par) half_o[d] = ot;
  __syncthreads();
  if (!par) out[h * 128 + d] = f32_to_bf16_bits((ot + half_o[d]) / lt);
}

__global__ void decode_advance_kernel(int64_t* state, int n) {
  for (int i = threadIdx.x; i < n; i += 256) state[i] += 1;
}

int gemv_grid(int iters, int cus) {
  const int cap = 8 * cus;
  if (iters <= cap) return iters;
  int best = cap, waste = -1;
  for (int g = cap; g >= cap / 2; --g) {                       // the grid size in [cap/2, cap] that wastes the fewest slots
    const int w = ((iters + g - 1) / g) * g - iters;
    if (waste < 0 || w < waste) { waste = w; best = g; }
    if (w == 0) break;
  }
  return best;
}

}  // namespace

int qp_decode_nsplit(const qp_ctx* ctx, int hkv) {
  int ns = 2 * ctx->cus / hkv;                                 // two workgroups per CU: one's loads under the other's arithmetic
  if (ns < 1) ns = 1;
  if (ns > 256) ns = 256;
  return ns;
}

size_t qp_decode_attn_workspace_bytes_impl(const qp_ctx* ctx, int hq, int hkv) {
  return (size_t)hq * (size_t)qp_decode_nsplit(ctx, hkv) * kDecPartial * sizeof(float);
}

int qp_launch_gemv(const qp_ctx* ctx, const void* w, const void* x, const void* norm_w, float eps, const void* bias, void* out,
                   int64_t n_out, int64_t k, int mode, hipStream_t s) {
  const int K16 = (int)(k / 8), N = (int)n_out;
  const bool pair = true;      // one row per wave-item (twice the workgroups for small N) measured the same per token (3.59 vs 3.60 ms)
  const int n_items = mode == GEMV_SWIGLU ? N : (pair ? (N + 1) / 2 : N);
  const int iters = (n_items + 3) / 4;
  const int grid = gemv_grid(iters, ctx->cus);
  const size_t shm = (size_t)k * 2;
  const float inv_hidden = 1.0f / (float)k;
  const uint4* W = (const uint4*)w; const uint4* X = (const uint4*)x; const uint4* NW = (const uint4*)norm_w;
  const uint16_t* B = (const uint16_t*)bias; uint16_t* O = (uint16_t*)out;
  if (mode == GEMV_SWIGLU) gemv_kernel<GEMV_SWIGLU, true><<<grid, 256, shm, s>>>(W, X, NW, nullptr, O, N, K16, iters, inv_hidden, eps);
  else if (mode == GEMV_BIAS && pair) gemv_kernel<GEMV_BIAS, true><<<grid, 256, shm, s>>>(W, X, NW, B, O, N, K16, iters, inv_hidden, eps);
  else if (mode == GEMV_BIAS) gemv_kernel<GEMV_BIAS, false><<<grid, 256, shm, s>>>(W, X, NW, B, O, N, K16, iters, inv_hidden, eps);
  else if (pair) gemv_kernel<GEMV_RESIDUAL, true><<<grid, 256, shm, s>>>(W, X, NW, nullptr, O, N, K16, iters, inv_hidden, eps);
  else gemv_kernel<GEMV_RESIDUAL, false><<<grid, 256, shm, s>>>(W, X, NW, nullptr, O, N, K16, iters, inv_hidden, eps);
  return qp_check_launch("gemv");
}

int qp_launch_decode_rope(const void* qkv, const int64_t* state, const void* cos_t, const void* sin_t, float theta, int hq, int hkv,
                          void* q_out, void* k_cache, void* v_cache, int64_t head_stride, hipStream_t s) {
  const int rows = hq + 2 * hkv;
  decode_rope_kernel<<<(rows + 15) / 16, 256, 0, s>>>((const uint4*)qkv, state, (const uint4*)cos_t, (const uint4*)sin_t, theta, hq,
                                                      hkv, (uint4*)q_out, (uint4*)k_cache,
                                                      (uint4*)v_cache, head_stride / 8);
  return qp_check_launch("decode_rope_append");
}

int qp_launch_decode_attn(const qp_ctx* ctx, const void* q, const void* k_cache, const void* v_cache, int64_t head_stride,
                          const int64_t* state, int hq, int hkv, float scale, void* out, void* workspace, hipStream_t s) {
  const int G = hq / hkv, ns = qp_decode_nsplit(ctx, hkv);
  const float c_log2 = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)ns, (unsigned)hkv);
  const uint4* Q = (const uint4*)q; const uint4* K = (const uint4*)k_cache; const uint4* V = (const uint4*)v_cache;
  float* ws = (float*)workspace;
  const bool use_valu = qp_dev().decode_attn_valu.load(std::memory_order_relaxed) != 0;   // developer A/B switch (qp_dev_switch)
  if (!use_valu && G <= 8) {
    decode_attn_mfma_kernel<false><<<grid, 256, 0, s>>>(Q, nullptr, nullptr, (uint4*)k_cache, (uint4*)v_cache, head_stride / 8, state,
                                                        hq, hkv, c_log2, ws);
    int rc = qp_check_launch("decode_attn");
    if (rc) return rc;
    decode_combine_kernel<<<hq, 256, 0, s>>>(ws, ns, c_log2, (uint16_t*)out);
    return qp_check_launch("decode_attn(combine)");
  }
  switch (G) {
    case 1: decode_attn_kernel<1><<<grid, 256, 0, s>>>(Q, K, V, head_stride / 8, state, c_log2, ws); break;
    case 2: decode_attn_kernel<2><<<grid, 256, 0, s>>>(Q, K, V, head_stride / 8, state, c_log2, ws); break;
    case 4: decode_attn_kernel<4><<<grid, 256, 0, s>>>(Q, K, V, head_stride / 8, state, c_log2, ws); break;
    case 6: decode_attn_kernel<6><<<grid, 256, 0, s>>>(Q, K, V, head_stride / 8, state, c_log2, ws); break;
    case 7: decode_attn_kernel<7><<<grid, 256, 0, s>>>(Q, K, V, head_stride / 8, state, c_log2, ws); break;
    case 8: decode_attn_kernel<8><<<grid, 256, 0, s>>>(Q, K, V, head_stride / 8, state, c_log2, ws); break;
    default: return qp_fail(QP_ERR_UNSUPPORTED, "qp_decode_attn: q heads per kv head must be 1, 2, 4, 6, 7 or 8 (got %d)", G);
  }
  int rc = qp_check_launch("decode_attn");
  if (rc) return rc;
  decode_combine_kernel<<<hq, 256, 0, s>>>(ws, ns, c_log2, (uint16_t*)out);
  return qp_check_launch("decode_attn(combine)");
}

int qp_launch_decode_attn_fused(const qp_ctx* ctx, const void* qkv, const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                                int64_t head_stride, const int64_t* state, int hq, int hkv, float scale, void* out, void* workspace,
                                hipStream_t s) {
  const int ns = qp_decode_nsplit(ctx, hkv);
  const float c_log2 = scale * 1.4426950408889634f;
  decode_attn_mfma_kernel<true><<<dim3((unsigned)ns, (unsigned)hkv), 256, 0, s>>>((const uint4*)qkv, (const uint4*)cos_t, (const uint4*)sin_t,
                                                                                 (uint4*)k_cache, (uint4*)v_cache, head_stride / 8, state,
                                                                                 hq, hkv, c_log2, (float*)workspace);
  int rc = qp_check_launch("decode_attn_fused");
  if (rc) return rc;
  decode_combine_kernel<<<hq, 256, 0, s>>>((const float*)workspace, ns, c_log2, (uint16_t*)out);
  return qp_check_launch("decode_attn_fused(combine)");
}

int qp_launch_decode_advance(int64_t* state, int n, hipStream_t s) {
  decode_advance_kernel<<<1, 256, 0, s>>>(state, n);
  return qp_check_launch("decode_advance");
}
