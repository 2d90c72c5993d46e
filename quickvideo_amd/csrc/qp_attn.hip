// Seam 3: prefill attention of one group over (pruned prefix, new tokens) — MFMA, gfx950.
// Reference: qwen25_lvu.py:61-62 (repeat_kv) + :102-112 (flash_attn causal, bottom-right aligned).
//
// Structure (wave64, v_mfma_f32_32x32x16_bf16; layouts verified on hardware by tools/probe/probe_layouts.hip):
//   workgroup = 4 waves = 128 query rows of ONE q head; each wave owns 32 query rows.
//   S^T = K.Q^T ("swapped" QK^T): A = K tile rows (keys), B = Q^T -> every lane holds 16 of the 32
//        keys of ONE query (lane&31), so the softmax row reduction is in-lane + one lane^32 exchange.
//   O^T = V^T.P : B = P straight from the S^T accumulator registers (the contraction order over keys is
//        permuted identically on both operands), A = V^T fetched with ds_read_b64_tr_b16 from a
//        [key/4][d/32][key%4][32] LDS image -> O^T accumulators keep query = lane, so the online-softmax
//        rescale is lane-local too.
//   K tile in LDS row-major [64][128] with the 16-B slot index XOR (row&15): conflict-free ds_read_b128.
//   KV is walked as two segments: prefix rows [0,P) (no causal mask) then the group's new rows (causal).
#include "qp_common.h"

namespace {

constexpr int kQB = 128;     // query rows per workgroup
constexpr int kKV = 64;      // keys per tile
constexpr int kD = 128;

struct AttnParams {
  const uint4* q; uint2* out;
  const uint4* kp; const uint4* vp; int64_t pre_hs16; int64_t P;
  const uint4* kn; const uint4* vn; int64_t new_hs16; int64_t n;
  int hq; int group; float c;   // c = scale * log2(e)
};

__device__ __forceinline__ bf16x8_t lds_read_b128(const unsigned char* lds, int off) {
  return *reinterpret_cast<const bf16x8_t*>(lds + off);
}

__device__ __forceinline__ s16x4_t lds_read_tr16(const unsigned char* lds, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + off));
}

// cooperative load of one 64-key K/V tile into LDS (256 threads, 4 x 16 B each per tensor)
__device__ __forceinline__ void load_tile(const uint4* __restrict__ ks, const uint4* __restrict__ vs, int64_t t0, int64_t seg_len,
                                          unsigned char* kl, unsigned char* vl, int tid) {
  const int slot = tid & 15;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = it * 16 + (tid >> 4);
    int64_t row = t0 + r;
    if (row >= seg_len) row = seg_len - 1;           // clamp: masked to -inf / multiplied by P = 0 later
    uint4 kv = ks[row * 16 + slot];
    uint4 vv = vs[row * 16 + slot];
    *reinterpret_cast<uint4*>(kl + r * 256 + ((slot ^ (r & 15)) << 4)) = kv;
    *reinterpret_cast<uint4*>(vl + (((r >> 2) * 4 + (slot >> 2)) << 8) + ((r & 3) << 6) + ((slot & 3) << 4)) = vv;
  }
}

template <bool kMask, bool kCausal>
__device__ __forceinline__ void tile_compute(const unsigned char* kl, const unsigned char* vl, const bf16x8_t (&qf)[8],
                                             f32x16_t (&o)[4], float& m_run, float& l_run, float c, int64_t t0,
                                             int64_t seg_len, int64_t qi, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  f32x16_t s[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    s[kb] = (f32x16_t){0};
    const int row = kb * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      bf16x8_t a = lds_read_b128(kl, row * 256 + (((kk * 2 + hi) ^ (row & 15)) << 4));
      s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s[kb], 0, 0, 0);
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (kMask) {
        const int64_t j = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const bool ok = (j < seg_len) && (!kCausal || j <= qi);
        s[kb][r] = ok ? s[kb][r] : -INFINITY;
      }
      mx = fmaxf(mx, s[kb][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(m_run, mx);
  const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
  const float mc = m_new * c;
  float rs = 0.f;
  bf16x8_t pf[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][cc * 8 + e], c, -mc));
        rs += p;
        pf[kb][cc][e] = (__bf16)p;
      }
  rs += __shfl_xor(rs, 32, 64);
  l_run = l_run * alpha + rs;
  m_run = m_new;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
  // O^T += V^T . P   (A = V^T fragment: lane -> d = db*32 + (lane&31), k-half = lane>>5)
  const int g1 = (lane >> 4) & 1;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int key0 = kb * 32 + cc * 16 + 4 * hi;       // keys key0..key0+3 and key0+8..key0+11
      const int kq = key0 >> 2;                           // 4-key row group
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int off = ((kq * 4 + db) << 8) + (((lane & 15) >> 2) << 6) + (g1 << 5) + ((lane & 3) << 3);
        s16x4_t v0 = lds_read_tr16(vl, off);
        s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));   // +8 keys = +2 row groups
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][cc], o[db], 0, 0, 0);
      }
    }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kKV * kD * 2];
  unsigned char* kl = lds;
  unsigned char* vl = lds + kKV * kD * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qb = gridDim.x - 1 - blockIdx.x;          // heaviest (latest) query blocks first
  const int head = blockIdx.y, kvh = head / p.group;
  const int64_t q0w = (int64_t)qb * kQB + wave * 32;
  const int64_t qi = q0w + (lane & 31);
  const int hi = lane >> 5;

  bf16x8_t qf[8];
  {
    const int64_t qrow = qi < p.n ? qi : p.n - 1;
    const uint4* qp = p.q + (qrow * p.hq + head) * 16;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = __builtin_bit_cast(bf16x8_t, qp[kk * 2 + hi]);
  }
  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x16_t){0};
  float m_run = -1e30f, l_run = 0.f;

  // ---- segment 1: pruned prefix, every key visible
  {
    const uint4* ks = p.kp + (int64_t)kvh * p.pre_hs16;
    const uint4* vs = p.vp + (int64_t)kvh * p.pre_hs16;
    for (int64_t t0 = 0; t0 < p.P; t0 += kKV) {
      __syncthreads();
      load_tile(ks, vs, t0, p.P, kl, vl, tid);
      __syncthreads();
      if (t0 + kKV <= p.P) tile_compute<false, false>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
      else tile_compute<true, false>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
    }
  }
  // ---- segment 2: the group's own keys, causal (key j visible to query i iff j <= i)
  {
    const uint4* ks = p.kn + (int64_t)kvh * p.new_hs16;
    const uint4* vs = p.vn + (int64_t)kvh * p.new_hs16;
    int64_t blk_end = (int64_t)qb * kQB + kQB;
    if (blk_end > p.n) blk_end = p.n;
    for (int64_t t0 = 0; t0 < blk_end; t0 += kKV) {
      __syncthreads();
      load_tile(ks, vs, t0, p.n, kl, vl, tid);
      __syncthreads();
      if (t0 <= q0w + 31) {                              // wave-uniform: some key of the tile is visible to this wave
        if (t0 + kKV - 1 <= q0w && t0 + kKV <= p.n) tile_compute<false, true>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
        else tile_compute<true, true>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
      }
    }
  }
  // ---- epilogue: O = O^T / l, bf16, 8-byte stores (4 consecutive d per register quad)
  if (qi < p.n) {
    const float inv = 1.0f / l_run;
    uint2* op = p.out + (qi * p.hq + head) * 32;       // 32 x 8 B per 128-wide row
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv), (__bf16)(o[db][r4 * 4 + 2] * inv),
                      (__bf16)(o[db][r4 * 4 + 3] * inv)};
        op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);     // d = db*32 + 8*r4 + 4*hi
      }
  }
}

}  // namespace

int qp_launch_prefill_attn(const qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix,
                           int64_t prefix_head_stride, int64_t prefix_len, const void* k_new, const void* v_new,
                           int64_t new_head_stride, int64_t n, int hq, int hkv, float scale, void* out, hipStream_t s) {
  (void)ctx;
  AttnParams p;
  p.q = (const uint4*)q; p.out = (uint2*)out;
  p.kp = (const uint4*)k_prefix; p.vp = (const uint4*)v_prefix; p.pre_hs16 = prefix_head_stride / 8; p.P = prefix_len;
  p.kn = (const uint4*)k_new; p.vn = (const uint4*)v_new; p.new_hs16 = new_head_stride / 8; p.n = n;
  p.hq = hq; p.group = hq / hkv; p.c = scale * 1.4426950408889634f;
  dim3 grid((unsigned)((n + kQB - 1) / kQB), (unsigned)hq);
  attn_fwd_kernel<<<grid, 256, 0, s>>>(p);
  return qp_check_launch("prefill_attn");
}
