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
#include <cstdlib>

namespace {

constexpr int kQB = 128;     // query rows per workgroup
constexpr int kKV = 64;      // keys per tile
constexpr int kD = 128;

struct AttnParams {
  const uint4* q; uint2* out;
  const uint4* kp; const uint4* vp; int64_t pre_hs16; int64_t P;
  const uint4* kn; const uint4* vn; int64_t new_hs16; int64_t n;
  int hq; int group; float c;   // c = scale * log2(e)
  int nqb; int hkv; int xcd_map;  // xcd_map: 1-D grid, workgroup b (XCD b%8) -> (kv head, q block) so an XCD's L2 serves one kv head
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


// ================================================================================================
// v2: LDS double buffer filled by direct global->LDS DMA (global_load_lds_dwordx4, swizzle applied on the
// per-lane SOURCE address, LDS image stays lane-linear), next tile in flight under the current tile's MFMAs,
// one barrier per tile; permlane32_swap for the cross-half reductions; deferred rescale (skip the O rescale
// while the running max grows by < 2^8); s_setprio around the MFMA clusters.
// ================================================================================================
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) unsigned char glb_u8;

__device__ __forceinline__ float xhalf_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <bool kMask, bool kCausal, bool kDefer, bool kPerm, bool kPrio>
__device__ __forceinline__ void tile_compute2(const unsigned char* kl, const unsigned char* vl, const bf16x8_t (&qf)[8],
                                              f32x16_t (&o)[4], float& m_run, float& l_run, float c, int64_t t0,
                                              int64_t seg_len, int64_t qi, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  f32x16_t s[2];
  if (kPrio) __builtin_amdgcn_s_setprio(1);
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
  if (kPrio) __builtin_amdgcn_s_setprio(0);
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
  mx = kPerm ? xhalf_max(mx) : fmaxf(mx, __shfl_xor(mx, 32, 64));
  // deferred rescale: keep the old reference max while it is within 2^8 of the tile max for EVERY row of the wave
  if (!kDefer || !__all((mx - m_run) * c <= 8.0f)) {
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
  l_run += kPerm ? xhalf_sum(rs) : rs + __shfl_xor(rs, 32, 64);
  const int g1 = (lane >> 4) & 1;
  const int voff = (((lane & 15) >> 2) << 6) + (g1 << 5) + ((lane & 3) << 3) + (hi << 10);
  if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int off = voff + (((kb * 8 + cc * 4) * 4 + db) << 8);     // row group kq = kb*8 + cc*4 + hi
        s16x4_t v0 = lds_read_tr16(vl, off);
        s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));
        typedef short s16x8_t __attribute__((ext_vector_type(8)));
        s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][cc], o[db], 0, 0, 0);
      }
    }
  if (kPrio) __builtin_amdgcn_s_setprio(0);
}

template <bool kDefer, bool kPerm, bool kPrio>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_v2(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kKV * kD * 2];     // [buf][K|V] 4 x 16 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = gridDim.x - 1 - blockIdx.x;
  const int head = blockIdx.y, kvh = head / p.group;
  const int64_t q0w = (int64_t)qb * kQB + wave * 32;
  const int64_t qi = q0w + (lane & 31);
  const int hi = lane >> 5;

  int64_t blk_end = (int64_t)qb * kQB + kQB;
  if (blk_end > p.n) blk_end = p.n;
  const int ntp = (int)((p.P + kKV - 1) / kKV), ntt = (int)((blk_end + kKV - 1) / kKV), nt = ntp + ntt;
  const uint4* kps = p.kp + (int64_t)kvh * p.pre_hs16;
  const uint4* vps = p.vp + (int64_t)kvh * p.pre_hs16;
  const uint4* kns = p.kn + (int64_t)kvh * p.new_hs16;
  const uint4* vns = p.vn + (int64_t)kvh * p.new_hs16;

  // per-lane constants of the DMA source addressing
  const int kr_in = lane >> 4;                        // K: row within the 4-row chunk
  const int vkey_in = (lane & 15) >> 2;               // V: key within the 4-key row group
  const int vslot = (lane >> 4) * 4 + (lane & 3);     // V: 16-B slot of the 256-B row

  auto issue = [&](int ti, int buf) {
    const bool pre = ti < ntp;
    const uint4* ks = pre ? kps : kns;
    const uint4* vs = pre ? vps : vns;
    const int64_t t0 = (int64_t)(pre ? ti : ti - ntp) * kKV;
    const int64_t last = (pre ? p.P : p.n) - 1;
    unsigned char* kb = lds + buf * 32768;
    unsigned char* vb = kb + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = wave * 4 + i;                     // 1-KB chunk of the tile (wave-uniform)
      const int r = 4 * c + kr_in;
      int64_t krow = t0 + r; if (krow > last) krow = last;
      const uint4* ksrc = ks + krow * 16 + ((lane & 15) ^ (r & 15));
      __builtin_amdgcn_global_load_lds((glb_u8*)ksrc, (lds_u8*)(kb + c * 1024), 16, 0, 0);
      int64_t vrow = t0 + 4 * c + vkey_in; if (vrow > last) vrow = last;
      const uint4* vsrc = vs + vrow * 16 + vslot;
      __builtin_amdgcn_global_load_lds((glb_u8*)vsrc, (lds_u8*)(vb + c * 1024), 16, 0, 0);
    }
  };

  issue(0, 0);
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int ti = 0; ti < nt; ++ti) {
    const int buf = ti & 1;
    if (ti + 1 < nt) issue(ti + 1, buf ^ 1);
    const unsigned char* kl = lds + buf * 32768;
    const unsigned char* vl = kl + 16384;
    if (ti < ntp) {
      const int64_t t0 = (int64_t)ti * kKV;
      if (t0 + kKV <= p.P) tile_compute2<false, false, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
      else tile_compute2<true, false, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
    } else {
      const int64_t t0 = (int64_t)(ti - ntp) * kKV;
      if (t0 <= q0w + 31) {
        if (t0 + kKV - 1 <= q0w && t0 + kKV <= p.n) tile_compute2<false, true, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
        else tile_compute2<true, true, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (qi < p.n) {
    const float inv = 1.0f / l_run;
    uint2* op = p.out + (qi * p.hq + head) * 32;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv), (__bf16)(o[db][r4 * 4 + 2] * inv),
                      (__bf16)(o[db][r4 * 4 + 3] * inv)};
        op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
      }
  }
}


// v3 family: v1's load scheme (register-staged, single LDS buffer, 2 workgroups per CU hide the latency)
// with the softmax-side refinements switchable for A/B.
template <bool kDefer, bool kPerm, bool kPrio>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_v3(AttnParams p) {
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
      if (t0 + kKV <= p.P) tile_compute2<false, false, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
      else tile_compute2<true, false, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
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
        if (t0 + kKV - 1 <= q0w && t0 + kKV <= p.n) tile_compute2<false, true, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
        else tile_compute2<true, true, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
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




// ================================================================================================
// w8: 8 waves (256 query rows) per workgroup, one workgroup per CU; K/V tiles fetched with buffer loads
// (wave-uniform tile offset in soffset, out-of-range rows read as zero -> no clamping / 64-bit address math),
// staged through registers into a double-buffered LDS image: loads of tile t+1 are issued before the MFMAs of
// tile t and written to the other buffer after them -> one barrier per tile, HBM/L2 latency hidden.
// ================================================================================================
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kQB8 = 256;

template <bool kDefer, bool kPerm, bool kPrio>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel_w8(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kKV * kD * 2];     // [buf][K|V] 4 x 16 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = gridDim.x - 1 - blockIdx.x;
  const int head = blockIdx.y, kvh = head / p.group;
  const int64_t q0w = (int64_t)qb * kQB8 + wave * 32;
  const int64_t qi = q0w + (lane & 31);
  const int hi = lane >> 5;

  int64_t blk_end = (int64_t)qb * kQB8 + kQB8;
  if (blk_end > p.n) blk_end = p.n;
  const int ntp = (int)((p.P + kKV - 1) / kKV), ntt = (int)((blk_end + kKV - 1) / kKV), nt = ntp + ntt;
  const __amdgpu_buffer_rsrc_t rkp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kp + (int64_t)kvh * p.pre_hs16), 0, (int)(p.P * 256), 0x00020000);
  const __amdgpu_buffer_rsrc_t rvp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vp + (int64_t)kvh * p.pre_hs16), 0, (int)(p.P * 256), 0x00020000);
  const __amdgpu_buffer_rsrc_t rkn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kn + (int64_t)kvh * p.new_hs16), 0, (int)(p.n * 256), 0x00020000);
  const __amdgpu_buffer_rsrc_t rvn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vn + (int64_t)kvh * p.new_hs16), 0, (int)(p.n * 256), 0x00020000);

  // per-thread constants: source byte offset inside a tile, LDS destinations (second chunk = +32 rows = +8192 B)
  const int r0 = tid >> 4, slot = tid & 15;
  const int src_off = r0 * 256 + slot * 16;
  const int kdst = r0 * 256 + ((slot ^ (r0 & 15)) << 4);
  const int vdst = (((r0 >> 2) * 4 + (slot >> 2)) << 8) + ((r0 & 3) << 6) + ((slot & 3) << 4);

  u32x4_t sk0, sk1, sv0, sv1;
  auto stage_load = [&](int ti) {
    const int soff = (ti < ntp ? ti : ti - ntp) * (kKV * 256);
    if (ti < ntp) {
      sk0 = __builtin_amdgcn_raw_buffer_load_b128(rkp, src_off, soff, 0);
      sk1 = __builtin_amdgcn_raw_buffer_load_b128(rkp, src_off + 8192, soff, 0);
      sv0 = __builtin_amdgcn_raw_buffer_load_b128(rvp, src_off, soff, 0);
      sv1 = __builtin_amdgcn_raw_buffer_load_b128(rvp, src_off + 8192, soff, 0);
    } else {
      sk0 = __builtin_amdgcn_raw_buffer_load_b128(rkn, src_off, soff, 0);
      sk1 = __builtin_amdgcn_raw_buffer_load_b128(rkn, src_off + 8192, soff, 0);
      sv0 = __builtin_amdgcn_raw_buffer_load_b128(rvn, src_off, soff, 0);
      sv1 = __builtin_amdgcn_raw_buffer_load_b128(rvn, src_off + 8192, soff, 0);
    }
  };
  auto stage_write = [&](int buf) {
    unsigned char* kb = lds + buf * 32768;
    unsigned char* vb = kb + 16384;
    *reinterpret_cast<u32x4_t*>(kb + kdst) = sk0;
    *reinterpret_cast<u32x4_t*>(kb + kdst + 8192) = sk1;
    *reinterpret_cast<u32x4_t*>(vb + vdst) = sv0;
    *reinterpret_cast<u32x4_t*>(vb + vdst + 8192) = sv1;
  };

  stage_load(0);
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
  stage_write(0);
  __syncthreads();

  for (int ti = 0; ti < nt; ++ti) {
    const int buf = ti & 1;
    const bool more = ti + 1 < nt;
    if (more) stage_load(ti + 1);
    const unsigned char* kl = lds + buf * 32768;
    const unsigned char* vl = kl + 16384;
    if (ti < ntp) {
      const int64_t t0 = (int64_t)ti * kKV;
      if (t0 + kKV <= p.P) tile_compute2<false, false, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
      else tile_compute2<true, false, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
    } else {
      const int64_t t0 = (int64_t)(ti - ntp) * kKV;
      if (t0 <= q0w + 31) {
        if (t0 + kKV - 1 <= q0w && t0 + kKV <= p.n) tile_compute2<false, true, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
        else tile_compute2<true, true, kDefer, kPerm, kPrio>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
      }
    }
    if (more) stage_write(buf ^ 1);
    __syncthreads();
  }
  if (qi < p.n) {
    const float inv = 1.0f / l_run;
    uint2* op = p.out + (qi * p.hq + head) * 32;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv), (__bf16)(o[db][r4 * 4 + 2] * inv),
                      (__bf16)(o[db][r4 * 4 + 3] * inv)};
        op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
      }
  }
}


// ================================================================================================
// s4: the v1 pipeline (single LDS buffer, 2 barriers per tile, 4 waves per workgroup, 2 workgroups per CU so one
// group's loads overlap the other's MFMAs) with ONE instance of the tile body in ONE loop over a unified tile
// index (prefix tiles then causal tail tiles) — keeps the O accumulators in fixed registers —, masking as a
// wave-uniform side branch, and buffer loads (tile offset in the scalar soffset, out-of-range rows read 0).
// ================================================================================================
template <bool kDefer, bool kPrio>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_s4(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kKV * kD * 2];
  unsigned char* kl = lds;
  unsigned char* vl = lds + kKV * kD * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int qb, head, kvh;
  if (p.xcd_map) {
    // Workgroup b is observed to run on XCD b % 8 (speed only, never correctness).  G = 8/Hkv XCDs serve one kv head:
    // all workgroups resident on an XCD stream the same K/V rows, so they hit in that XCD's private L2.
    const int G = 8 / p.hkv, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    kvh = xcd / G;
    const int j = slot * G + (xcd % G);                  // item of this kv head: heaviest (latest) q blocks first
    if (j >= p.nqb * p.group) return;
    qb = p.nqb - 1 - j / p.group;
    head = kvh * p.group + j % p.group;
  } else {
    qb = gridDim.x - 1 - blockIdx.x;
    head = blockIdx.y; kvh = head / p.group;
  }
  const int q0w = qb * kQB + wave * 32;
  const int qi = q0w + (lane & 31);
  const int hi = lane >> 5, l31 = lane & 31;
  const int n = (int)p.n, P = (int)p.P;

  int blk_end = qb * kQB + kQB;
  if (blk_end > n) blk_end = n;
  const int ntp = (P + kKV - 1) / kKV, ntt = (blk_end + kKV - 1) / kKV, nt = ntp + ntt;
  const __amdgpu_buffer_rsrc_t rkp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kp + (int64_t)kvh * p.pre_hs16), 0, P * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vp + (int64_t)kvh * p.pre_hs16), 0, P * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rkn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kn + (int64_t)kvh * p.new_hs16), 0, n * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vn + (int64_t)kvh * p.new_hs16), 0, n * 256, 0x00020000);

  const int r0 = tid >> 4, slot = tid & 15;              // this thread's rows r0 + 16*it, 16-B slot
  const int src_off = r0 * 256 + slot * 16;
  const int kdst = r0 * 256 + ((slot ^ (r0 & 15)) << 4);                                         // + it*4096
  const int vdst = (((r0 >> 2) * 4 + (slot >> 2)) << 8) + ((r0 & 3) << 6) + ((slot & 3) << 4);   // + it*4096
  // LDS read addresses (tile independent)
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = l31 * 256 + (((kk * 2 + hi) ^ (l31 & 15)) << 4);       // + kb*8192
  const int voff = (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3) + (hi << 10);

  bf16x8_t qf[8];
  {
    const int qrow = qi < n ? qi : n - 1;
    const uint4* qp = p.q + ((int64_t)qrow * p.hq + head) * 16;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = __builtin_bit_cast(bf16x8_t, qp[kk * 2 + hi]);
  }
  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x16_t){0};
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.c;

  for (int ti = 0; ti < nt; ++ti) {
    const bool pre = ti < ntp;
    const int t0 = (pre ? ti : ti - ntp) * kKV;
    const int seg_len = pre ? P : n;
    {
      const int soff = t0 * 256;
      u32x4_t kv[4], vv[4];
      if (pre) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          kv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkp, src_off + it * 4096, soff, 0);
          vv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvp, src_off + it * 4096, soff, 0);
        }
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          kv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkn, src_off + it * 4096, soff, 0);
          vv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvn, src_off + it * 4096, soff, 0);
        }
      }
      __syncthreads();                                  // everyone is done reading the previous tile
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        *reinterpret_cast<u32x4_t*>(kl + kdst + it * 4096) = kv[it];
        *reinterpret_cast<u32x4_t*>(vl + vdst + it * 4096) = vv[it];
      }
      __syncthreads();
    }
    if (!pre && t0 > q0w + 31) continue;                // wave-uniform: no key of this tile is visible to this wave
    f32x16_t s[2];
    if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = (f32x16_t){0};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        bf16x8_t a = lds_read_b128(kl, koff[kk] + kb * 8192);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s[kb], 0, 0, 0);
      }
    }
    if (kPrio) __builtin_amdgcn_s_setprio(0);
    const bool need_mask = (t0 + kKV > seg_len) || (!pre && t0 + kKV - 1 > q0w);
    if (need_mask) {                                    // wave-uniform
      const int qlim = pre ? 0x7fffffff : qi;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s[kb][r] = (j < seg_len && j <= qlim) ? s[kb][r] : -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = xhalf_max(mx);
    if (!kDefer || !__all((mx - m_run) * c <= 8.0f)) {
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
    bf16x8_t pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][cc * 8 + e], c, -mc));
          rs += pe;
          pf[kb][cc][e] = (__bf16)pe;
        }
    l_run += xhalf_sum(rs);
    if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int off = voff + (((kb * 8 + cc * 4) * 4 + db) << 8);
          s16x4_t v0 = lds_read_tr16(vl, off);
          s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));
          typedef short s16x8_t __attribute__((ext_vector_type(8)));
          s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][cc], o[db], 0, 0, 0);
        }
      }
    if (kPrio) __builtin_amdgcn_s_setprio(0);
  }
  if (qi < n) {
    const float inv = 1.0f / l_run;
    uint2* op = p.out + ((int64_t)qi * p.hq + head) * 32;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv), (__bf16)(o[db][r4 * 4 + 2] * inv),
                      (__bf16)(o[db][r4 * 4 + 3] * inv)};
        op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
      }
  }
}


// d4: s4 with the next tile's buffer loads issued BEFORE the current tile's MFMAs (register-staged prefetch);
// kDbuf = second LDS buffer, staged registers written after the MFMAs, one barrier per tile.
template <bool kDefer, bool kPrio, bool kDbuf>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_d4(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[(kDbuf ? 4 : 2) * kKV * kD * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = gridDim.x - 1 - blockIdx.x;
  const int head = blockIdx.y, kvh = head / p.group;
  const int q0w = qb * kQB + wave * 32;
  const int qi = q0w + (lane & 31);
  const int hi = lane >> 5, l31 = lane & 31;
  const int n = (int)p.n, P = (int)p.P;

  int blk_end = qb * kQB + kQB;
  if (blk_end > n) blk_end = n;
  const int ntp = (P + kKV - 1) / kKV, ntt = (blk_end + kKV - 1) / kKV, nt = ntp + ntt;
  const __amdgpu_buffer_rsrc_t rkp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kp + (int64_t)kvh * p.pre_hs16), 0, P * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vp + (int64_t)kvh * p.pre_hs16), 0, P * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rkn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kn + (int64_t)kvh * p.new_hs16), 0, n * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vn + (int64_t)kvh * p.new_hs16), 0, n * 256, 0x00020000);

  const int r0 = tid >> 4, slot = tid & 15;              // this thread's rows r0 + 16*it, 16-B slot
  const int src_off = r0 * 256 + slot * 16;
  const int kdst = r0 * 256 + ((slot ^ (r0 & 15)) << 4);                                         // + it*4096
  const int vdst = (((r0 >> 2) * 4 + (slot >> 2)) << 8) + ((r0 & 3) << 6) + ((slot & 3) << 4);   // + it*4096
  // LDS read addresses (tile independent)
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = l31 * 256 + (((kk * 2 + hi) ^ (l31 & 15)) << 4);       // + kb*8192
  const int voff = (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3) + (hi << 10);

  bf16x8_t qf[8];
  {
    const int qrow = qi < n ? qi : n - 1;
    const uint4* qp = p.q + ((int64_t)qrow * p.hq + head) * 16;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = __builtin_bit_cast(bf16x8_t, qp[kk * 2 + hi]);
  }
  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x16_t){0};
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.c;

  u32x4_t kv[4], vv[4];
  auto stage_load = [&](int ti) {
    const bool pre_ = ti < ntp;
    const int soff = (pre_ ? ti : ti - ntp) * (kKV * 256);
    if (pre_) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        kv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkp, src_off + it * 4096, soff, 0);
        vv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvp, src_off + it * 4096, soff, 0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        kv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkn, src_off + it * 4096, soff, 0);
        vv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvn, src_off + it * 4096, soff, 0);
      }
    }
  };
  auto stage_write = [&](unsigned char* kb_, unsigned char* vb_) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      *reinterpret_cast<u32x4_t*>(kb_ + kdst + it * 4096) = kv[it];
      *reinterpret_cast<u32x4_t*>(vb_ + vdst + it * 4096) = vv[it];
    }
  };
  stage_load(0);
  if (kDbuf) { stage_write(lds, lds + 16384); __syncthreads(); }

  for (int ti = 0; ti < nt; ++ti) {
    const bool pre = ti < ntp;
    const int t0 = (pre ? ti : ti - ntp) * kKV;
    const int seg_len = pre ? P : n;
    const unsigned char* kl = lds + (kDbuf ? (ti & 1) * 32768 : 0);
    const unsigned char* vl = kl + 16384;
    if (!kDbuf) {
      __syncthreads();                                  // everyone is done reading the previous tile
      stage_write(lds, lds + 16384);
      __syncthreads();
    }
    if (ti + 1 < nt) stage_load(ti + 1);                // in flight under this tile's MFMAs
    if (pre || t0 <= q0w + 31) {                        // wave-uniform: some key of this tile is visible to this wave
    f32x16_t s[2];
    if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = (f32x16_t){0};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        bf16x8_t a = lds_read_b128(kl, koff[kk] + kb * 8192);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s[kb], 0, 0, 0);
      }
    }
    if (kPrio) __builtin_amdgcn_s_setprio(0);
    const bool need_mask = (t0 + kKV > seg_len) || (!pre && t0 + kKV - 1 > q0w);
    if (need_mask) {                                    // wave-uniform
      const int qlim = pre ? 0x7fffffff : qi;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s[kb][r] = (j < seg_len && j <= qlim) ? s[kb][r] : -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = xhalf_max(mx);
    if (!kDefer || !__all((mx - m_run) * c <= 8.0f)) {
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
    bf16x8_t pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][cc * 8 + e], c, -mc));
          rs += pe;
          pf[kb][cc][e] = (__bf16)pe;
        }
    l_run += xhalf_sum(rs);
    if (kPrio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int off = voff + (((kb * 8 + cc * 4) * 4 + db) << 8);
          s16x4_t v0 = lds_read_tr16(vl, off);
          s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));
          typedef short s16x8_t __attribute__((ext_vector_type(8)));
          s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][cc], o[db], 0, 0, 0);
        }
      }
    if (kPrio) __builtin_amdgcn_s_setprio(0);
    }
    if (kDbuf) {
      if (ti + 1 < nt) stage_write(lds + ((ti + 1) & 1) * 32768, lds + ((ti + 1) & 1) * 32768 + 16384);
      __syncthreads();
    }
  }
  if (qi < n) {
    const float inv = 1.0f / l_run;
    uint2* op = p.out + ((int64_t)qi * p.hq + head) * 32;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv), (__bf16)(o[db][r4 * 4 + 2] * inv),
                      (__bf16)(o[db][r4 * 4 + 3] * inv)};
        op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
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
  p.nqb = (int)((n + kQB - 1) / kQB); p.hkv = hkv; p.xcd_map = 0;
  dim3 grid((unsigned)((n + kQB - 1) / kQB), (unsigned)hq);
  const char* var = getenv("QP_ATTN_VARIANT");        // developer A/B switch; default = newest validated variant
  const int variant = var ? atoi(var) : 10;
  switch (variant) {
    case 1: attn_fwd_kernel<<<grid, 256, 0, s>>>(p); break;
    case 3: attn_fwd_kernel_v3<false, true, false><<<grid, 256, 0, s>>>(p); break;
    case 4: attn_fwd_kernel_v3<true, true, false><<<grid, 256, 0, s>>>(p); break;
    case 5: attn_fwd_kernel_v3<true, true, true><<<grid, 256, 0, s>>>(p); break;
    case 6: attn_fwd_kernel_v2<false, false, false><<<grid, 256, 0, s>>>(p); break;
    case 7: attn_fwd_kernel_v3<true, false, false><<<grid, 256, 0, s>>>(p); break;
    case 12: case 13: {
      if (prefix_len * 256 >= (1ll << 31) || n * 256 >= (1ll << 31)) { attn_fwd_kernel_v3<true, true, true><<<grid, 256, 0, s>>>(p); break; }
      if (variant == 12) attn_fwd_kernel_d4<true, true, false><<<grid, 256, 0, s>>>(p);
      else attn_fwd_kernel_d4<true, true, true><<<grid, 256, 0, s>>>(p);
      break;
    }
    case 10: case 11: {
      if (prefix_len * 256 >= (1ll << 31) || n * 256 >= (1ll << 31)) { attn_fwd_kernel_v3<true, true, true><<<grid, 256, 0, s>>>(p); break; }
      if (variant == 10 && hkv <= 8 && 8 % hkv == 0) {
        p.xcd_map = 1;
        const int G = 8 / hkv, items = p.nqb * p.group;
        attn_fwd_kernel_s4<true, true><<<dim3(8 * ((items + G - 1) / G)), 256, 0, s>>>(p);
      } else attn_fwd_kernel_s4<true, true><<<grid, 256, 0, s>>>(p);
      break;
    }
    case 8: case 9: {
      if (prefix_len * 256 >= (1ll << 31) || n * 256 >= (1ll << 31)) { attn_fwd_kernel_v3<true, true, true><<<grid, 256, 0, s>>>(p); break; }
      dim3 g8((unsigned)((n + kQB8 - 1) / kQB8), (unsigned)hq);
      if (variant == 8) attn_fwd_kernel_w8<true, true, true><<<g8, 512, 0, s>>>(p);
      else attn_fwd_kernel_w8<true, true, false><<<g8, 512, 0, s>>>(p);
      break;
    }
    default: attn_fwd_kernel_v2<true, true, true><<<grid, 256, 0, s>>>(p); break;
  }
  return qp_check_launch("prefill_attn");
}
