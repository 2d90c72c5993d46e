// Glue kernels of the patched decoder layer (qwen25_lvu.py:166-198): residual add + RMSNorm, SwiGLU.
// HBM-bound, 16-B vector accesses; math mirrors transformers' Qwen2RMSNorm / Qwen2MLP on bf16 tensors
// (fp32 inside, one bf16 rounding per torch op).
#include "qp_common.h"

__global__ __launch_bounds__(256) void add_rmsnorm_kernel(uint4* __restrict__ h, const uint4* __restrict__ delta,
                                                          const uint4* __restrict__ w, uint4* __restrict__ out, int hidden16,
                                                          float inv_hidden, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* row = (uint4*)smem;                       // hidden16 chunks of the (updated) row
  __shared__ float red[4];
  const int64_t base = (int64_t)blockIdx.x * hidden16;
  float ss = 0.f;
  for (int c = threadIdx.x; c < hidden16; c += 256) {
    uint4 x = h[base + c];
    if (delta) {
      uint4 d = delta[base + c];
      unsigned xw[4] = {x.x, x.y, x.z, x.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float lo = __uint_as_float(xw[i] << 16) + __uint_as_float(dw[i] << 16);
        float hi = __uint_as_float(xw[i] & 0xffff0000u) + __uint_as_float(dw[i] & 0xffff0000u);
        xw[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
      }
      x = make_uint4(xw[0], xw[1], xw[2], xw[3]);
      h[base + c] = x;
    }
    row[c] = x;
    unsigned xw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo = __uint_as_float(xw[i] << 16), hi = __uint_as_float(xw[i] & 0xffff0000u);
      ss = __builtin_fmaf(lo, lo, ss);
      ss = __builtin_fmaf(hi, hi, ss);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = (red[0] + red[1]) + (red[2] + red[3]);
  const float rs = 1.0f / __fsqrt_rn(tot * inv_hidden + eps);
  for (int c = threadIdx.x; c < hidden16; c += 256) {
    uint4 x = row[c], ww = w[c];
    unsigned xw[4] = {x.x, x.y, x.z, x.w}, wv[4] = {ww.x, ww.y, ww.z, ww.w}, o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo = round_bf16(__uint_as_float(xw[i] << 16) * rs) * __uint_as_float(wv[i] << 16);
      float hi = round_bf16(__uint_as_float(xw[i] & 0xffff0000u) * rs) * __uint_as_float(wv[i] & 0xffff0000u);
      o[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
    }
    out[base + c] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

int qp_launch_add_rmsnorm(void* h, const void* delta, const void* w, void* out, int64_t n, int hidden, float eps,
                          hipStream_t s) {
  if (n == 0) return QP_OK;
  add_rmsnorm_kernel<<<(int)n, 256, (size_t)hidden * 2, s>>>((uint4*)h, (const uint4*)delta, (const uint4*)w, (uint4*)out,
                                                              hidden / 8, 1.0f / (float)hidden, eps);
  return qp_check_launch("add_rmsnorm");
}

__global__ __launch_bounds__(256) void add_inplace_kernel(uint4* __restrict__ h, const uint4* __restrict__ delta, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    uint4 x = h[i], d = delta[i];
    unsigned xw[4] = {x.x, x.y, x.z, x.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float lo = __uint_as_float(xw[k] << 16) + __uint_as_float(dw[k] << 16);
      float hi = __uint_as_float(xw[k] & 0xffff0000u) + __uint_as_float(dw[k] & 0xffff0000u);
      xw[k] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
    }
    h[i] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
  }
}

int qp_launch_add_inplace(void* h, const void* delta, int64_t n_elems, hipStream_t s) {
  int64_t n16 = n_elems / 8;
  int64_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  add_inplace_kernel<<<(int)blocks, 256, 0, s>>>((uint4*)h, (const uint4*)delta, n16);
  return qp_check_launch("add_inplace");
}

__device__ __forceinline__ float silu_mul_bf16(float g, float u) {
  float a = round_bf16(g / (1.0f + __expf(-g)));
  return a * u;
}

// gate / up rows of `row16` uint4 each (one fused [n][2*inter] projection: up = gate + inter16, row16 = 2*inter16; two separate
// [n][inter] projections: row16 = inter16)
__global__ __launch_bounds__(256) void swiglu_kernel(const uint4* __restrict__ gate, const uint4* __restrict__ up, int64_t row16,
                                                     int64_t n, int inter16, uint4* __restrict__ out) {
  const int64_t total = n * inter16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / inter16, c = i - t * inter16;
    uint4 g = gate[t * row16 + c], u = up[t * row16 + c];
    unsigned gw[4] = {g.x, g.y, g.z, g.w}, uw[4] = {u.x, u.y, u.z, u.w}, o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float lo = silu_mul_bf16(__uint_as_float(gw[k] << 16), __uint_as_float(uw[k] << 16));
      float hi = silu_mul_bf16(__uint_as_float(gw[k] & 0xffff0000u), __uint_as_float(uw[k] & 0xffff0000u));
      o[k] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

int qp_launch_swiglu(const void* gate, const void* up, int64_t row_elems, int64_t n, int inter, void* out, hipStream_t s) {
  int64_t total = n * (inter / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (blocks < 1) blocks = 1;
  swiglu_kernel<<<(int)blocks, 256, 0, s>>>((const uint4*)gate, (const uint4*)up, row_elems / 8, n, inter / 8, (uint4*)out);
  return qp_check_launch("swiglu");
}

// ------------------------------------------------------------------------------------------------
// ViT front-end glue (transformers Qwen2-VL vision tower [3P]): 2-D rotary on q,k of the fused qkv projection and
// quick-GELU.  qkv bf16 [n][3][H][hd]; cos/sin fp32 [n][hd/2] (the tower's rotary table; both halves of a head use
// the same angles).  Like apply_rotary_pos_emb_vision the rotation is done in fp32 and rounded once.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_rope_kernel(uint4* __restrict__ qkv, const float* __restrict__ cos_t,
                                                       const float* __restrict__ sin_t, int64_t n, int heads, int half8) {
  // one thread per (token, q|k, head, 8-element chunk of the first half); partner chunk = +half elements
  const int64_t per_tok = 2ll * heads * half8;
  const int64_t total = n * per_tok;
  const int row16 = 3 * heads * half8 * 2;                    // uint4 per token row
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per_tok;
    int r = (int)(i - t * per_tok);
    const int c = r % half8; r /= half8;
    const int h = r % heads; const int which = r / heads;      // 0 = q, 1 = k
    const int64_t base = t * row16 + ((int64_t)which * heads + h) * (2 * half8) + c;
    uint4 a = qkv[base], b = qkv[base + half8];
    unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, ao[4], bo[4];
    const float* cs = cos_t + t * (half8 * 8) + c * 8;
    const float* sn = sin_t + t * (half8 * 8) + c * 8;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      // hipcc compiles with -ffp-contract=fast: nothing but this pragma keeps the multiplies and the add from becoming v_pk_fma_f32 (they did, the moment the bf16 conversion became one instruction and the
      // loop body got vectorised; the result then differs from torch's three separately rounded ops in the last bit)
#pragma clang fp contract(off)
      float x0 = __uint_as_float(aw[w] << 16), x1 = __uint_as_float(aw[w] & 0xffff0000u);
      float y0 = __uint_as_float(bw[w] << 16), y1 = __uint_as_float(bw[w] & 0xffff0000u);
      float c0 = cs[2 * w], c1 = cs[2 * w + 1], s0 = sn[2 * w], s1 = sn[2 * w + 1];
      // first half: x*cos + (-y)*sin ; second half: y*cos + x*sin
      // separately rounded fp32 mul / mul / add like the torch ops (no fma contraction)
      // (plain * and + written HERE: the contract flag of an operation is the one of the scope it is written in, so the header's
      // __fmul_rn / __fadd_rn — inline functions compiled under contract=fast — would still fuse)
      const float p00 = x0 * c0, p01 = (-y0) * s0, p10 = x1 * c1, p11 = (-y1) * s1;
      const float q00 = y0 * c0, q01 = x0 * s0, q10 = y1 * c1, q11 = x1 * s1;
      ao[w] = (unsigned)f32_to_bf16_bits(p00 + p01) | ((unsigned)f32_to_bf16_bits(p10 + p11) << 16);
      bo[w] = (unsigned)f32_to_bf16_bits(q00 + q01) | ((unsigned)f32_to_bf16_bits(q10 + q11) << 16);
    }
    qkv[base] = make_uint4(ao[0], ao[1], ao[2], ao[3]);
    qkv[base + half8] = make_uint4(bo[0], bo[1], bo[2], bo[3]);
  }
}

int qp_launch_vit_rope(void* qkv, const float* cos_t, const float* sin_t, int64_t n, int heads, int head_dim, hipStream_t s) {
  const int half8 = head_dim / 16;
  int64_t total = n * 2 * heads * half8;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  vit_rope_kernel<<<(int)blocks, 256, 0, s>>>((uint4*)qkv, cos_t, sin_t, n, heads, half8);
  return qp_check_launch("vit_rope");
}

// out = bf16( y * bf16(sigmoid(bf16(1.702*y))) )  — torch's `y * torch.sigmoid(1.702 * y)` on bf16 tensors
__global__ __launch_bounds__(256) void quick_gelu_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    uint4 v = x[i];
    unsigned w[4] = {v.x, v.y, v.z, v.w}, o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xffff0000u);
      float tl = round_bf16(1.702f * lo), th = round_bf16(1.702f * hi);
      float sl = round_bf16(1.0f / (1.0f + __expf(-tl))), sh = round_bf16(1.0f / (1.0f + __expf(-th)));
      o[k] = (unsigned)f32_to_bf16_bits(lo * sl) | ((unsigned)f32_to_bf16_bits(hi * sh) << 16);
    }
    out[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

int qp_launch_quick_gelu(const void* x, void* out, int64_t n_elems, hipStream_t s) {
  int64_t n16 = n_elems / 8;
  int64_t blocks = (n16 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  quick_gelu_kernel<<<(int)blocks, 256, 0, s>>>((const uint4*)x, (uint4*)out, n16);
  return qp_check_launch("quick_gelu");
}

// Front end, first step: uint8 frames [F, 3, H, W] -> pixel rows [gt*gh*gw][row_elems] bf16 in the HF Qwen2VLImageProcessor order
// (t, h/mg, w/mg, mg, mg | C, tp, ps, ps) [3P], rescaled + CLIP-normalised, in ONE pass.  The reference does this on the CPU under the GIL (HF
// processor, qwen25_lvu_interleaved.py:252-271, 318-340) and uploads 4 bytes per value; rounds 1-5 did it on the GPU with five torch passes.
// The arithmetic is a 3 x 256-entry table: lut[c][v] = bf16((v * (1/255) - mean[c]) / std[c]) is computed ONCE by the caller with the very
// expression the torch path uses, so the kernel is a pure gather — bit-identical by construction.  Columns [C*tp*ps*ps, row_elems) are
// written as zeros: the patch-embedding GEMM then runs on a tile-aligned K (1176 -> 1280: 46 -> 28 us per group of the 1-hour video).
__global__ __launch_bounds__(256) void patchify_kernel(const unsigned char* __restrict__ frames, const unsigned short* __restrict__ lut,
                                                       unsigned short* __restrict__ out, int H, int W, int gh, int gw, int ps, int tp, int mg,
                                                       int cols, int row_elems, int64_t n_rows) {
  __shared__ unsigned short s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  const int groups = row_elems / 8;                                 // 8 output values (16 bytes) per thread
  const int pp = ps * ps, cpp = tp * pp;
  const int uw = gw / mg, uh = gh / mg, mm = mg * mg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rows * groups; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / groups;
    const int c0 = (int)(i % groups) * 8;
    const int unit = (int)(r / mm), sub = (int)(r % mm);
    const int wb = unit % uw, t2 = unit / uw, hb = t2 % uh, t = t2 / uh;
    const int h = hb * mg + sub / mg, w = wb * mg + sub % mg;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = c0 + e;
      if (col < cols) {
        const int c = col / cpp, rem = col % cpp, tt = rem / pp, rem2 = rem % pp, py = rem2 / ps, px = rem2 % ps;
        const int64_t src = (((int64_t)(t * tp + tt) * 3 + c) * H + (h * ps + py)) * W + (w * ps + px);
        v[e] = s_lut[c * 256 + frames[src]];
      } else {
        v[e] = 0;
      }
    }
    uint4 o;
    o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16); o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
    reinterpret_cast<uint4*>(out)[i] = o;
  }
}

int qp_launch_patchify(const void* frames, const void* lut, void* out, int n_frames, int H, int W, int ps, int tp, int mg, int row_elems, hipStream_t s) {
  const int gt = n_frames / tp, gh = H / ps, gw = W / ps, cols = 3 * tp * ps * ps;
  const int64_t n_rows = (int64_t)gt * gh * gw, work = n_rows * (row_elems / 8);
  int64_t blocks = (work + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (blocks < 1) blocks = 1;
  patchify_kernel<<<(int)blocks, 256, 0, s>>>((const unsigned char*)frames, (const unsigned short*)lut, (unsigned short*)out, H, W, gh, gw, ps, tp, mg,
                                              cols, row_elems, n_rows);
  return qp_check_launch("patchify");
}

// x = bf16(x + delta) (when delta != NULL, written back);  out = bf16((x - mean) * rstd * w + b)  — the residual add of a ViT
// block fused with the LayerNorm that follows it (transformers Qwen2VLVisionBlock [3P]: x = x + attn(norm1(x)); x = x + mlp(norm2(x))).
// One WAVE per row, the row stays in registers (hidden <= 4096): two-pass mean / variance in fp32 like torch's layer_norm.
__global__ __launch_bounds__(256) void add_layernorm_kernel(uint4* __restrict__ x, const uint4* __restrict__ delta,
                                                            const uint4* __restrict__ w, const uint4* __restrict__ b,
                                                            uint4* __restrict__ out, int64_t n, int hidden16, float inv_hidden,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int64_t base = row * hidden16;
  float v[8][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int idx = lane + 64 * c;
    if (idx < hidden16) {
      uint4 xv = x[base + idx];
      unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
      if (delta) {
        const uint4 dv = delta[base + idx];
        const unsigned dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = __uint_as_float(xw[i] << 16) + __uint_as_float(dw[i] << 16);
          const float hi = __uint_as_float(xw[i] & 0xffff0000u) + __uint_as_float(dw[i] & 0xffff0000u);
          xw[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
        }
        x[base + idx] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[c][2 * i] = __uint_as_float(xw[i] << 16); v[c][2 * i + 1] = __uint_as_float(xw[i] & 0xffff0000u);
        sum += v[c][2 * i] + v[c][2 * i + 1];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum * inv_hidden;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    if (lane + 64 * c < hidden16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; sq = __builtin_fmaf(d, d, sq); }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / __fsqrt_rn(sq * inv_hidden + eps);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int idx = lane + 64 * c;
    if (idx < hidden16) {
      const uint4 wv = w[idx], bv = b[idx];
      const unsigned ww[4] = {wv.x, wv.y, wv.z, wv.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
      unsigned o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = (v[c][2 * i] - mean) * rstd * __uint_as_float(ww[i] << 16) + __uint_as_float(bw[i] << 16);
        const float hi = (v[c][2 * i + 1] - mean) * rstd * __uint_as_float(ww[i] & 0xffff0000u) + __uint_as_float(bw[i] & 0xffff0000u);
        o[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
      }
      out[base + idx] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

int qp_launch_add_layernorm(void* x, const void* delta, const void* w, const void* b, void* out, int64_t n, int hidden, float eps,
                            hipStream_t s) {
  if (n == 0) return QP_OK;
  add_layernorm_kernel<<<(unsigned)((n + 3) / 4), 256, 0, s>>>((uint4*)x, (const uint4*)delta, (const uint4*)w, (const uint4*)b,
                                                               (uint4*)out, n, hidden / 8, 1.0f / (float)hidden, eps);
  return qp_check_launch("add_layernorm");
}
