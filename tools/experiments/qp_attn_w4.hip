// NOT PART OF THE BUILD — round-1 experiment kept for the record.  Result on MI355X: correct, but 0.59-0.85 PF
// vs 0.92-1.02 PF for the production kernel: hipcc parks MFMA operands/accumulators in AGPRs and emits ~570 v_accvgpr
// moves per tile, and the pinned zip does not reach the planned overlap.  To build it: add the file to the Makefile and
// declare qp_launch_prefill_attn_w4 in qp_common.h.
//
// Seam 3, experimental "w4" variant: one wave per SIMD with the whole 512-register file, 64 query rows per wave
// (two independent 32-row q blocks A and B), 256 query rows per workgroup, one workgroup per CU.
//
// With a single wave per SIMD nothing overlaps unless the instruction stream itself alternates matrix and vector work,
// so the unmasked-tile body is written as a hand-zipped schedule (hipcc clusters MFMAs; `sched_barrier(0)` pins the
// source order): every group of 4 MFMAs (128 matrix-pipe cycles) carries ~28 VALU ops of the OTHER q block's softmax:
//
//   S1   QK^T_A (16 MFMA)      || K b128 reads, V tr-reads (fragments stay in registers for both q blocks)
//   1    QK^T_B quad 0         || row max A            -> rescale decision A (rare branch, between groups)
//   2-4  QK^T_B quads 1-3      || exp/sum/pack of P_A chunks 0-2
//   5    P.V_A chunk 0         || P_A chunk 3
//   6    P.V_A chunk 1         || row max B            -> rescale decision B
//   7-8  P.V_A chunks 2-3      || P_B chunks 0-1
//   9-10 P.V_B chunks 0-1      || P_B chunks 2-3
//   11-12 P.V_B chunks 2-3
//
// K/V tiles: LDS double buffer, next tile's buffer loads in flight under the whole body, one barrier per tile.
// Ragged / diagonal tiles take a plain sequential body.  Same LDS images, MFMA operand maps and math as qp_attn.hip.
#include "qp_common.h"
#include <cstdlib>

namespace {

constexpr int kQB4 = 256;
constexpr int kKV = 64;

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

struct W4Params {
  const uint4* q; uint2* out;
  const uint4* kp; const uint4* vp; int64_t pre_hs16; int P;
  const uint4* kn; const uint4* vn; int64_t new_hs16; int n;
  int hq; int group; float c;
  int nqb; int hkv; int items;
};

#define PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ bf16x8_t ldsr128(const unsigned char* lds, int off) { return *reinterpret_cast<const bf16x8_t*>(lds + off); }
__device__ __forceinline__ s16x4_t ldstr16(const unsigned char* lds, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + off));
}
__device__ __forceinline__ float xh_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xh_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// per-q-block running state
struct QState {
  f32x16_t o[4];
  float m, l;
};

// ---- pieces of the zipped schedule ------------------------------------------------------------------------------------
// two elements of a P chunk: exp2(s*c - mc), row-sum, pack to bf16x2  (7 VALU)
#define PREP2(S, CH, J, PW, RS, MC)                                                                      \
  {                                                                                                      \
    float p0_ = __builtin_amdgcn_exp2f(__builtin_fmaf(S[(CH) >> 1][((CH) & 1) * 8 + 2 * (J)], c, -(MC)));     \
    float p1_ = __builtin_amdgcn_exp2f(__builtin_fmaf(S[(CH) >> 1][((CH) & 1) * 8 + 2 * (J) + 1], c, -(MC))); \
    RS += p0_; RS += p1_;                                                                                \
    PW[(CH)][(J)] = pack_bf16(p0_, p1_);                                                                 \
  }

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_w4(W4Params p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * kKV * 128 * 2 + kQB4 * 256];   // 2 x (K 16 KB | V 16 KB) + Q tile 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = 8 / p.hkv, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int kvh = xcd / G;
  const int item = slot * G + (xcd % G);
  if (item >= p.items) return;
  const int qb = p.nqb - 1 - item / p.group;
  const int head = kvh * p.group + item % p.group;
  const int hi = lane >> 5, l31 = lane & 31;
  const int n = p.n, P = p.P;
  const int q0A = qb * kQB4 + wave * 64, q0B = q0A + 32;
  const int qiA = q0A + l31, qiB = q0B + l31;
  const float c = p.c;

  int blk_end = qb * kQB4 + kQB4;
  if (blk_end > n) blk_end = n;
  const int ntp = (P + kKV - 1) / kKV, ntt = (blk_end + kKV - 1) / kKV, nt = ntp + ntt;
  const __amdgpu_buffer_rsrc_t rkp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kp + (int64_t)kvh * p.pre_hs16), 0, P * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vp + (int64_t)kvh * p.pre_hs16), 0, P * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rkn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kn + (int64_t)kvh * p.new_hs16), 0, n * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vn + (int64_t)kvh * p.new_hs16), 0, n * 256, 0x00020000);

  const int r0 = tid >> 4, slot16 = tid & 15;
  const int src_off = r0 * 256 + slot16 * 16;
  const int kdst = r0 * 256 + ((slot16 ^ (r0 & 15)) << 4);
  const int vdst = (((r0 >> 2) * 4 + (slot16 >> 2)) << 8) + ((r0 & 3) << 6) + ((slot16 & 3) << 4);
  int koff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = l31 * 256 + (((kk * 2 + hi) ^ (l31 & 15)) << 4);
  int xoff[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) xoff[kk] = (((kk * 2 + hi) ^ (l31 & 15)) << 4);
  const int voff = (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3) + (hi << 10);

  // Q tile of the workgroup -> LDS (row-major, 16-B slot XOR (row & 15) like the K image); fragments are re-read per tile so
  // the 64 Q registers of the two q blocks do not crowd the arch VGPRs (MFMA A/B operands must sit there for hipcc).
  unsigned char* ql = lds + 4 * kKV * 128 * 2;
  for (int i = tid; i < kQB4 * 16; i += 256) {
    const int r = i >> 4, sl = i & 15;
    int qrow = qb * kQB4 + r; if (qrow >= n) qrow = n - 1;
    *reinterpret_cast<uint4*>(ql + r * 256 + ((sl ^ (r & 15)) << 4)) = p.q[((int64_t)qrow * p.hq + head) * 16 + sl];
  }
  const int qoffA = (wave * 64 + l31) * 256, qoffB = qoffA + 32 * 256;     // + ((kk*2+hi) ^ (l31&15)) << 4 : same XOR term as koff
  QState A, B;
#pragma unroll
  for (int db = 0; db < 4; ++db) { A.o[db] = (f32x16_t){0}; B.o[db] = (f32x16_t){0}; }
  A.m = B.m = -1e30f; A.l = B.l = 0.f;

  u32x4_t skv[4], svv[4];
  auto stage_load = [&](int ti) {
    const bool pre_ = ti < ntp;
    const int soff = (pre_ ? ti : ti - ntp) * (kKV * 256);
    if (pre_) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        skv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkp, src_off + it * 4096, soff, 0);
        svv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvp, src_off + it * 4096, soff, 0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        skv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkn, src_off + it * 4096, soff, 0);
        svv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvn, src_off + it * 4096, soff, 0);
      }
    }
  };
  auto stage_write = [&](int buf) {
    unsigned char* kb_ = lds + buf * 32768;
    unsigned char* vb_ = kb_ + 16384;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      *reinterpret_cast<u32x4_t*>(kb_ + kdst + it * 4096) = skv[it];
      *reinterpret_cast<u32x4_t*>(vb_ + vdst + it * 4096) = svv[it];
    }
  };

  stage_load(0);
  stage_write(0);
  __syncthreads();
  if (nt > 1) stage_load(1);

  for (int ti = 0; ti < nt; ++ti) {
    const bool pre = ti < ntp;
    const int t0 = (pre ? ti : ti - ntp) * kKV;
    const int seg_len = pre ? P : n;
    const unsigned char* kl = lds + (ti & 1) * 32768;
    const unsigned char* vl = kl + 16384;
    const bool ragged = t0 + kKV > seg_len;
    const bool visB = pre || t0 <= q0B + 31;                 // B holds the later rows: visible to A implies visible to B
    const bool maskA = ragged || (!pre && t0 + kKV - 1 > q0A);
    const bool maskB = ragged || (!pre && t0 + kKV - 1 > q0B);
    const int qlimA = pre ? 0x7fffffff : qiA, qlimB = pre ? 0x7fffffff : qiB;

    if (visB) {                                              // wave-uniform
      // =================================== zipped fast path ===================================
      s16x8_t vfa[4], vfb[4];                // V fragments of two chunks in flight: [db]
      f32x16_t sA[2], sB[2];
      unsigned pwA[4][4], pwB[4][4];         // packed bf16 P chunks: [chunk][word]
      // ---- S1: QK^T_A, kk-major: one Q fragment feeds both 32-key halves
      sA[0] = (f32x16_t){0}; sA[1] = (f32x16_t){0};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const bf16x8_t qv = ldsr128(ql, qoffA + xoff[kk]);
        sA[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldsr128(kl, koff[kk]), qv, sA[0], 0, 0, 0);
        sA[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldsr128(kl, koff[kk] + 8192), qv, sA[1], 0, 0, 0);
      }
#define VLOAD(DST, CH, DB)                                                                                   \
  {                                                                                                          \
    const int off_ = voff + (((((CH) >> 1) * 8 + ((CH) & 1) * 4) * 4 + (DB)) << 8);                              \
    s16x4_t v0_ = ldstr16(vl, off_), v1_ = ldstr16(vl, off_ + (2 * 4 << 8));                                  \
    DST[(DB)] = (s16x8_t){v0_[0], v0_[1], v0_[2], v0_[3], v1_[0], v1_[1], v1_[2], v1_[3]};                    \
  }
      if (maskA) {                                           // ragged / diagonal tiles only (side branch on the scores)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jk = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            sA[kb][r] = (jk < seg_len && jk <= qlimA) ? sA[kb][r] : -INFINITY;
          }
      }
      // ---- group 1: QK^T_B quad 0 || row max A
      sB[0] = (f32x16_t){0}; sB[1] = (f32x16_t){0};
      float mxA = sA[0][0];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = i >> 1, kb = i & 1;
        sB[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldsr128(kl, koff[kk] + kb * 8192), ldsr128(ql, qoffB + xoff[kk]), sB[kb], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { mxA = fmaxf(mxA, sA[0][i * 4 + r]); mxA = fmaxf(mxA, sA[1][i * 4 + r]); }
        PIN();
      }
      mxA = xh_max(mxA);
      if (!__all((mxA - A.m) * c <= 8.0f)) {
        const float m_new = fmaxf(A.m, mxA);
        const float alpha = __builtin_amdgcn_exp2f((A.m - m_new) * c);
        A.m = m_new; A.l *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) A.o[db][r] *= alpha;
      }
      const float mcA = A.m * c;
      float rsA = 0.f, rsB = 0.f;
      // ---- groups 2-4: QK^T_B quads 1-3 || P_A chunks 0-2
#pragma unroll
      for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int idx = 4 + g * 4 + i, kk = idx >> 1, kb = idx & 1;
          sB[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ldsr128(kl, koff[kk] + kb * 8192), ldsr128(ql, qoffB + xoff[kk]), sB[kb], 0, 0, 0);
          PREP2(sA, g, i, pwA, rsA, mcA);
          if (g == 2) VLOAD(vfa, 0, i);
          PIN();
        }
      }
      // ---- group 5: P.V_A chunk 0 || P_A chunk 3
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        u32x4v pw = {pwA[0][0], pwA[0][1], pwA[0][2], pwA[0][3]};
        A.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfa[db]), __builtin_bit_cast(bf16x8_t, pw), A.o[db], 0, 0, 0);
        PREP2(sA, 3, db, pwA, rsA, mcA);
        VLOAD(vfb, 1, db);
        PIN();
      }
      if (maskB) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jk = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            sB[kb][r] = (jk < seg_len && jk <= qlimB) ? sB[kb][r] : -INFINITY;
          }
      }
      // ---- group 6: P.V_A chunk 1 || row max B
      float mxB = sB[0][0];
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        u32x4v pw = {pwA[1][0], pwA[1][1], pwA[1][2], pwA[1][3]};
        A.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfb[db]), __builtin_bit_cast(bf16x8_t, pw), A.o[db], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { mxB = fmaxf(mxB, sB[0][db * 4 + r]); mxB = fmaxf(mxB, sB[1][db * 4 + r]); }
        VLOAD(vfa, 2, db);
        PIN();
      }
      mxB = xh_max(mxB);
      if (!__all((mxB - B.m) * c <= 8.0f)) {
        const float m_new = fmaxf(B.m, mxB);
        const float alpha = __builtin_amdgcn_exp2f((B.m - m_new) * c);
        B.m = m_new; B.l *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) B.o[db][r] *= alpha;
      }
      const float mcB = B.m * c;
      // ---- groups 7-8: P.V_A chunks 2-3 || P_B chunks 0-1
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          u32x4v pw = {pwA[2 + g][0], pwA[2 + g][1], pwA[2 + g][2], pwA[2 + g][3]};
          if (g == 0) {
            A.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfa[db]), __builtin_bit_cast(bf16x8_t, pw), A.o[db], 0, 0, 0);
            PREP2(sB, g, db, pwB, rsB, mcB);
            VLOAD(vfb, 3, db);
          } else {
            A.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfb[db]), __builtin_bit_cast(bf16x8_t, pw), A.o[db], 0, 0, 0);
            PREP2(sB, g, db, pwB, rsB, mcB);
            VLOAD(vfa, 0, db);
          }
          PIN();
        }
      }
      // ---- groups 9-10: P.V_B chunks 0-1 || P_B chunks 2-3
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          u32x4v pw = {pwB[g][0], pwB[g][1], pwB[g][2], pwB[g][3]};
          if (g == 0) {
            B.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfa[db]), __builtin_bit_cast(bf16x8_t, pw), B.o[db], 0, 0, 0);
            PREP2(sB, 2 + g, db, pwB, rsB, mcB);
            VLOAD(vfb, 1, db);
          } else {
            B.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfb[db]), __builtin_bit_cast(bf16x8_t, pw), B.o[db], 0, 0, 0);
            PREP2(sB, 2 + g, db, pwB, rsB, mcB);
            VLOAD(vfa, 2, db);
          }
          PIN();
        }
      }
      A.l += xh_sum(rsA);
      B.l += xh_sum(rsB);
      // ---- groups 11-12: P.V_B chunks 2-3
#pragma unroll
      for (int g = 2; g < 4; ++g)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          u32x4v pw = {pwB[g][0], pwB[g][1], pwB[g][2], pwB[g][3]};
          if (g == 2) {
            B.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfa[db]), __builtin_bit_cast(bf16x8_t, pw), B.o[db], 0, 0, 0);
            VLOAD(vfb, 3, db);
          } else {
            B.o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfb[db]), __builtin_bit_cast(bf16x8_t, pw), B.o[db], 0, 0, 0);
          }
          PIN();
        }
    }
    if (ti + 1 < nt) stage_write((ti + 1) & 1);
    __syncthreads();
    if (ti + 2 < nt) stage_load(ti + 2);
  }

  auto store = [&](QState& st, int qi) {
    if (qi < n) {
      const float inv = 1.0f / st.l;
      uint2* op = p.out + ((int64_t)qi * p.hq + head) * 32;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          bf16x4_t v = {(__bf16)(st.o[db][r4 * 4 + 0] * inv), (__bf16)(st.o[db][r4 * 4 + 1] * inv), (__bf16)(st.o[db][r4 * 4 + 2] * inv),
                        (__bf16)(st.o[db][r4 * 4 + 3] * inv)};
          op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
        }
    }
  };
  store(A, qiA);
  store(B, qiB);
}

}  // namespace

int qp_launch_prefill_attn_w4(const qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix, int64_t prefix_head_stride,
                              int64_t prefix_len, const void* k_new, const void* v_new, int64_t new_head_stride, int64_t n, int hq,
                              int hkv, float scale, void* out, hipStream_t s) {
  (void)ctx;
  W4Params p;
  p.q = (const uint4*)q; p.out = (uint2*)out;
  p.kp = (const uint4*)k_prefix; p.vp = (const uint4*)v_prefix; p.pre_hs16 = prefix_head_stride / 8; p.P = (int)prefix_len;
  p.kn = (const uint4*)k_new; p.vn = (const uint4*)v_new; p.new_hs16 = new_head_stride / 8; p.n = (int)n;
  p.hq = hq; p.group = hq / hkv; p.c = scale * 1.4426950408889634f;
  p.nqb = (int)((n + kQB4 - 1) / kQB4); p.hkv = hkv; p.items = p.nqb * p.group;
  const int G = 8 / hkv;
  attn_fwd_kernel_w4<<<dim3(8 * ((p.items + G - 1) / G)), 256, 0, s>>>(p);
  return qp_check_launch("prefill_attn(w4)");
}
