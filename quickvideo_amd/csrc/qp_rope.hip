// Seam 2 kernels: M-RoPE table + fused rotate / KV append / key sum-of-squares.
// Reference: qwen25_lvu.py:46-58 (view, apply_multimodal_rotary_pos_emb, cache.update) and
// transformers' Qwen2VLRotaryEmbedding (fp32 angles, cos/sin rounded to the model dtype = bf16).
#include "qp_common.h"

// cos/sin[t][i] for frequency index i < D/2: angle = pos[section(i)][t] * theta^(-2i/D), fp32, rounded to bf16.
// section(i): first s0 indices use the temporal stream, next s1 height, next s2 width (mrope_section).
__global__ __launch_bounds__(256) void mrope_table_kernel(const int64_t* __restrict__ pos, int64_t n, int s0, int s1,
                                                          float theta, int half, uint16_t* __restrict__ cos_out,
                                                          uint16_t* __restrict__ sin_out) {
  const int64_t total = n * half;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / half;
    const int f = (int)(i - t * half);
    const int sec = f < s0 ? 0 : (f < s0 + s1 ? 1 : 2);
    // inv_freq = 1 / theta^(2f/D), computed like torch: base ** (arange(0,D,2)/D) in fp32
    const float expo = (float)(2 * f) / (float)(2 * half);
    const float inv_freq = 1.0f / powf(theta, expo);
    const float ang = (float)pos[sec * n + t] * inv_freq;
    cos_out[i] = f32_to_bf16_bits(cosf(ang));
    sin_out[i] = f32_to_bf16_bits(sinf(ang));
  }
}

int qp_launch_mrope_table(const int64_t* pos, int64_t n, const int32_t* sections, float theta, int head_dim, void* cos_out,
                          void* sin_out, hipStream_t s) {
  int64_t total = n * (head_dim / 2);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  mrope_table_kernel<<<(int)blocks, 256, 0, s>>>(pos, n, sections[0], sections[1], theta, head_dim / 2, (uint16_t*)cos_out,
                                                 (uint16_t*)sin_out);
  return qp_check_launch("mrope_table");
}

// One 16-lane group per head row of 128 bf16 (lane c owns elements 8c..8c+7; rotate-half partner = lane c^8).
// Rows of a token: n_q query heads, n_kv key heads, n_kv value heads, in the order of the fused projection.
__device__ __forceinline__ float sumsq8(const unsigned short o[8]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { float x = bf16_bits_to_f32(o[e]); s = __builtin_fmaf(x, x, s); }
  return s;
}

// kKeys: additionally reduce the per-head sums of a token to its 16-bit norm key (heads added in ascending order, correctly
// rounded sqrt, bf16 RNE: bit-identical to select_kernel / the oracle) for prune_keys_kernel (qp_prune.hip).  Requires the hkv
// key rows of a token to sit in ONE wave: the launcher checks (qp_rope_can_fuse_keys), otherwise norm_keys_kernel does this
// step from head_sumsq.
template <bool kKeys>
__global__ __launch_bounds__(256) void rope_append_kernel(const uint4* __restrict__ qkv, const uint4* __restrict__ cos_t,
                                                          const uint4* __restrict__ sin_t, int64_t n, int hq, int hkv,
                                                          uint4* __restrict__ q_out, uint4* __restrict__ k_dst,
                                                          uint4* __restrict__ v_dst, int64_t dst_hs16, int64_t dst_row0,
                                                          float* __restrict__ head_sumsq, uint16_t* __restrict__ norm_keys,
                                                          int largest) {
  const int c = threadIdx.x & 15;
  const int rows_per_tok = hq + 2 * hkv;
  const int64_t rows = n * rows_per_tok;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
    const int64_t t = r / rows_per_tok;
    const int hr = (int)(r - t * rows_per_tok);
    uint4 x = qkv[r * 16 + c];
    if (hr >= hq + hkv) {   // value row: straight copy into the arena / staging block
      v_dst[(int64_t)(hr - hq - hkv) * dst_hs16 + (dst_row0 + t) * 16 + c] = x;
      continue;             // (16-lane groups are uniform in hr, so the shuffles below stay convergent per group)
    }
    uint4 cs = cos_t[t * 8 + (c & 7)], sn = sin_t[t * 8 + (c & 7)];
    uint4 p;                // partner chunk (elements +-64)
    p.x = __shfl_xor((int)x.x, 8, 16); p.y = __shfl_xor((int)x.y, 8, 16);
    p.z = __shfl_xor((int)x.z, 8, 16); p.w = __shfl_xor((int)x.w, 8, 16);
    const float sign = (c < 8) ? -1.f : 1.f;   // rotate_half: first half gets -x[i+64], second half +x[i-64]
    unsigned xw[4] = {x.x, x.y, x.z, x.w}, pw[4] = {p.x, p.y, p.z, p.w}, cw[4] = {cs.x, cs.y, cs.z, cs.w},
             sw[4] = {sn.x, sn.y, sn.z, sn.w};
    unsigned short o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int w = e >> 1, sh = (e & 1) * 16;
      float xv = bf16_bits_to_f32((unsigned short)(xw[w] >> sh)), pv = bf16_bits_to_f32((unsigned short)(pw[w] >> sh));
      float cv = bf16_bits_to_f32((unsigned short)(cw[w] >> sh)), sv = bf16_bits_to_f32((unsigned short)(sw[w] >> sh));
      float a = round_bf16(xv * cv);
      float b = round_bf16((sign * pv) * sv);
      o[e] = f32_to_bf16_bits(a + b);
    }
    uint4 ov;
    ov.x = o[0] | ((unsigned)o[1] << 16); ov.y = o[2] | ((unsigned)o[3] << 16);
    ov.z = o[4] | ((unsigned)o[5] << 16); ov.w = o[6] | ((unsigned)o[7] << 16);
    if (hr < hq) {
      q_out[(t * hq + hr) * 16 + c] = ov;
    } else {
      const int h = hr - hq;
      k_dst[(int64_t)h * dst_hs16 + (dst_row0 + t) * 16 + c] = ov;
      if (head_sumsq || kKeys) {
        float s = sumsq8(o);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) s = s + __shfl_xor(s, m, 16);
        if (c == 0 && head_sumsq) head_sumsq[(int64_t)h * n + t] = s;
        if (kKeys) {
          // the token's key rows are the 16-lane groups g - h ... g - h + hkv - 1 of this wave (g = this group's index)
          const int g0 = ((threadIdx.x & 63) >> 4) - h;
          float tot = __shfl(s, g0 * 16, 64);
          for (int j = 1; j < hkv; ++j) tot = tot + __shfl(s, (g0 + j) * 16, 64);
          if (h == 0 && c == 0) {
            const unsigned short b = f32_to_bf16_bits(sqrt_rn_f32(tot));
            norm_keys[t] = largest ? (unsigned short)~b : b;
          }
        }
      }
    }
  }
}

// the hkv key rows of every token lie inside one 4-row wave (rows are dealt 16 per workgroup pass, 4 per wave, in row order)
bool qp_rope_can_fuse_keys(int hq, int hkv) {
  const int rpt = hq + 2 * hkv;
  return (hkv == 1) || (hkv == 2 && rpt % 2 == 0 && hq % 2 == 0) || (hkv == 4 && rpt % 4 == 0 && hq % 4 == 0);
}

int qp_launch_rope_append(const void* qkv, const void* cos, const void* sin, int64_t n, int hq, int hkv, void* q_out,
                          void* k_dst, void* v_dst, int64_t dst_head_stride, int64_t dst_row0, float* head_sumsq,
                          uint16_t* norm_keys, int largest, hipStream_t s) {
  int64_t rows = n * (hq + 2 * hkv);
  int64_t blocks = (rows + 15) / 16;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (norm_keys)
    rope_append_kernel<true><<<(int)blocks, 256, 0, s>>>((const uint4*)qkv, (const uint4*)cos, (const uint4*)sin, n, hq, hkv,
                                                         (uint4*)q_out, (uint4*)k_dst, (uint4*)v_dst, dst_head_stride / 8, dst_row0,
                                                         head_sumsq, norm_keys, largest);
  else
    rope_append_kernel<false><<<(int)blocks, 256, 0, s>>>((const uint4*)qkv, (const uint4*)cos, (const uint4*)sin, n, hq, hkv,
                                                          (uint4*)q_out, (uint4*)k_dst, (uint4*)v_dst, dst_head_stride / 8, dst_row0,
                                                          head_sumsq, nullptr, 0);
  return qp_check_launch("rope_append");
}
