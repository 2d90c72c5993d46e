// Seam 3, "s7" form of the software-pipelined prefill attention: ONE wave per SIMD, 64 query rows per wave.
// Same math, layouts, work items (256-row items, kv-split partials, combine) and LDS tile images as qp_attn_s6.hip; what changes:
//   * a wave owns two 32-row q blocks A and B, so every K / V fragment it reads from LDS feeds two MFMAs: half the LDS reads,
//     counted waits, LDS-DMA pieces and barriers per MFMA (on real data the kernel is power-limited, not schedule-limited:
//     energy per FLOP is what is left to gain);
//   * that needs 2 x 64 accumulator registers for O plus two score sets per q block: more than the 256 architectural VGPRs, so
//     O (and the Q fragments) live in AGPRs.  hipcc's own placement of builtin MFMAs shuffled ~570 registers per tile through
//     v_accvgpr_* in this shape (qp_attn_w4.hip); here the MFMAs are inline asm whose constraints ("+a" for O, "a"
//     for Q, "v" for scores / K / V / P) pin the register files (tools/probe/probe_agpr_accum.hip).
// hipcc cannot see into the asm MFMAs, so nothing reads an MFMA result early: scores are consumed at least 24 gaps later, and
// the three places that touch them right away (mask branch, rescale branch, epilogue) start with an explicit s_nop fence.
// Step schedule + counted LDS waits: tools/gen_attn_s7.py -> qp_attn_s7_iter.inc.
#include "qp_attn.h"

using namespace qpattn;

namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#define PIN() __builtin_amdgcn_sched_barrier(0)
#define KEEP(X) asm volatile("" : "+v"(X))
#define DSR_B128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DSR_TR16(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define S7_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory")
#define S7_KREAD(SLOT, KK, IMM) DSR_B128(kr[SLOT], koffv[KK], IMM)
#define S7_VREAD(SLOT, ADDR, O0, O1) { DSR_TR16(vr[SLOT][0], ADDR, O0); DSR_TR16(vr[SLOT][1], ADDR, O1); }
// The MFMAs themselves are literal asm statements emitted by the generator: scores in VGPRs (operands), O and the Q fragments in
// FIXED AGPRs that hipcc never sees as variables (a[0:63] = O of q block A, a[64:127] = O of B, a[128:159] / a[160:191] = Q of A / B;
// every statement lists the AGPRs it writes as clobbers).  The kernel needs no VGPR spill, so hipcc has no use for AGPRs itself.
#define S7_PV_OPERANDS(SLOT, PW, C)                                                                                  \
  const s16x4_t v0_ = vr[SLOT][0], v1_ = vr[SLOT][1];                                                                \
  const s16x8_t av_ = {v0_[0], v0_[1], v0_[2], v0_[3], v1_[0], v1_[1], v1_[2], v1_[3]};                              \
  const u32x4v pc_ = {PW[4 * (C)], PW[4 * (C) + 1], PW[4 * (C) + 2], PW[4 * (C) + 3]}
// fence before ordinary code reads registers an asm MFMA has just written (hipcc does not know the hazard)
#define S7_FENCE_S(S) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S##A[0]), "+v"(S##A[1]), "+v"(S##B[0]), "+v"(S##B[1]))
#define S7_FENCE_O() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
// softmax element pipeline, one per q block (ST = A | B)
#define S7_ELA(ST, S, E, XI) xs##ST[XI] = __builtin_fmaf(S[(E) >> 4][(E) & 15], c, nmc##ST)
#define S7_ELB(ST, XI, PI) ps##ST[PI] = __builtin_amdgcn_exp2f(xs##ST[XI])
#define S7_ELC(ST, PI) rs##ST += ps##ST[PI]
#define S7_PACK(ST, W, PA, PB) { pw##ST[W] = pack_bf16(ps##ST[PA], ps##ST[PB]); KEEP(pw##ST[W]); }
#define S7_ELKEEP() asm volatile("" : "+v"(xsA[0]), "+v"(xsA[1]), "+v"(psA[0]), "+v"(psA[1]), "+v"(psA[2]), "+v"(psA[3]), "+v"(rsA), \
                                 "+v"(xsB[0]), "+v"(xsB[1]), "+v"(psB[0]), "+v"(psB[1]), "+v"(psB[2]), "+v"(psB[3]), "+v"(rsB))
#define S7_DMA(RSRC, LDSOFF, VOFF, SOFF) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, (__attribute__((address_space(3))) void*)(lds3 + (LDSOFF)), 16, VOFF, SOFF, 0, 0)
#define S7_DMA_K(K, LDSBASE) S7_DMA(rk2, (LDSBASE) + (K) * 4096 + wave * 1024, ksrc_off + (K) * 4096, soff_k2)
#define S7_DMA_V(K) S7_DMA(rv1, vdma_off + (K) * 4096 + wave * 1024, vsrc_off + (K) * 4096, soff_v1)
#define S7_MAX4(ST, S, Q)                                                               \
  {                                                                                     \
    if ((Q) == 0) mx##ST = S[0][0];                                                     \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) mx##ST = fmaxf(mx##ST, S[(Q) >> 1][((Q) & 1) * 8 + e_]); \
    KEEP(mx##ST);                                                                       \
  }
#define S7_MASK1(S, QLIM)                                                               \
  _Pragma("unroll") for (int kb_ = 0; kb_ < 2; ++kb_)                                    \
    _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                  \
      const int jk_ = t0_n + kb_ * 32 + (r_ & 3) + 8 * (r_ >> 2) + 4 * hi;               \
      S[kb_][r_] = (jk_ < seg_n && jk_ <= (QLIM)) ? S[kb_][r_] : -INFINITY;              \
    }
#define S7_MASK_NEXT(S)                                                                 \
  if (mask_n) {                                                                         \
    S7_FENCE_S(S);                                                                      \
    S7_MASK1(S##A, qlim_nA)                                                             \
    S7_MASK1(S##B, qlim_nB)                                                             \
  }

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

constexpr int kKB0 = 0, kKB1 = 16384, kVB = 32768;   // LDS: K0 | K1 | V0 | V1 (16 KB each)
constexpr int kRows7 = 256;                          // query rows per workgroup: 4 waves x (A: 32 rows | B: 32 rows)

template <bool kXcd>
__global__ __launch_bounds__(256, 1) void attn_fwd_kernel_s7(AttnParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
  const unsigned lds32 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int kvh, j;
  if (kXcd) {
    const int G = 8 / p.hkv, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    kvh = xcd / G;
    j = slot * G + (xcd % G);
  } else {
    kvh = blockIdx.y; j = blockIdx.x;
  }
  int item, split = 0;
  const bool partial = j >= p.n_whole;
  if (!partial) item = j;
  else { const int r = j - p.n_whole; item = p.n_whole + r / p.nsplit; split = r % p.nsplit; }
  if (item >= p.items) return;
  const int qb = p.nqb - 1 - item / p.group;
  const int head = kvh * p.group + item % p.group;
  const int hi = lane >> 5, l31 = lane & 31;
  const int q0lA = qb * kRows7 + wave * 64, q0lB = q0lA + 32;      // local rows of q blocks A and B
  const int qiA = q0lA + l31, qiB = q0lB + l31;
  const int q0wA = p.q_row0 + q0lA;                                // rows inside the group's new segment (causal mask); B = A + 32
  const int n = (int)p.n, P = (int)p.P, nq = p.nq;

  int blk_end = qb * kRows7 + kRows7;
  if (blk_end > nq) blk_end = nq;
  blk_end += p.q_row0;
  const int ntp = (P + kKV - 1) / kKV, ntt = (blk_end + kKV - 1) / kKV, nt = ntp + ntt;
  int ti_lo = 0, ti_hi = nt;
  if (partial) { ti_lo = (int)((int64_t)split * nt / p.nsplit); ti_hi = (int)((int64_t)(split + 1) * nt / p.nsplit); }
  const uint4* kp_base = p.kp + (int64_t)kvh * p.pre_hs16;
  const uint4* vp_base = p.vp + (int64_t)kvh * p.pre_hs16;
  const uint4* kn_base = p.kn + (int64_t)kvh * p.new_hs16;
  const uint4* vn_base = p.vn + (int64_t)kvh * p.new_hs16;
  auto k_rsrc = [&](int tj) {
    const bool pre_ = tj < ntp;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(pre_ ? kp_base : kn_base), 0, (pre_ ? P : n) * 256, 0x00020000);
  };
  auto v_rsrc = [&](int tj) {
    const bool pre_ = tj < ntp;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(pre_ ? vp_base : vn_base), 0, (pre_ ? P : n) * 256, 0x00020000);
  };
  auto tile_soff = [&](int tj) { return (tj < ntp ? tj : tj - ntp) * (kKV * 256); };

  __attribute__((address_space(3))) unsigned char* lds3 = (__attribute__((address_space(3))) unsigned char*)lds;
  const int r0 = tid >> 4, slot16 = tid & 15;
  const int ksrc_off = r0 * 256 + ((slot16 ^ (r0 & 15)) << 4);                                                         // + it*4096
  const int vsrc_off = ((r0 & ~3) + ((lane >> 2) & 3)) * 256 + ((((lane >> 4) << 2) | (lane & 3)) << 4);               // + it*4096
  unsigned koffv[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koffv[kk] = lds32 + l31 * 256 + (((kk * 2 + hi) ^ (l31 & 15)) << 4);                  // + kb*8192
  const unsigned voffv = lds32 + (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3) + (hi << 10);

  {
    const int rowA = qiA < nq ? qiA : nq - 1, rowB = qiB < nq ? qiB : nq - 1;
    const uint4* qap = p.q + ((int64_t)rowA * p.hq + head) * 16;
    const uint4* qbp = p.q + ((int64_t)rowB * p.hq + head) * 16;
    uint4 qa[8], qb[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { qa[kk] = qap[kk * 2 + hi]; qb[kk] = qbp[kk * 2 + hi]; }
#define S7_PART 4
#include "qp_attn_s7_iter.inc"
#undef S7_PART
  }
  float m_runA = -1e30f, m_runB = -1e30f;
  const float c = p.c;

  // ---- prologue: K(lo), V(lo), K(lo+1) to LDS ----------------------------------------------------------------------------
  {
    const __amdgpu_buffer_rsrc_t rk0 = k_rsrc(ti_lo), rv0 = v_rsrc(ti_lo), rk1 = k_rsrc(ti_lo + 1);
    const int so0 = tile_soff(ti_lo), so1 = tile_soff(ti_lo + 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      S7_DMA(rk0, kKB0 + it * 4096 + wave * 1024, ksrc_off + it * 4096, so0);
      S7_DMA(rv0, kVB + it * 4096 + wave * 1024, vsrc_off + it * 4096, so0);
      S7_DMA(rk1, kKB1 + it * 4096 + wave * 1024, ksrc_off + it * 4096, so1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x16_t s0A[2], s0B[2], s1A[2], s1B[2];
  float rsA = 0.f, rsB = 0.f, mxA, mxB, nmcA, nmcB;
  float xsA[2] = {0.f, 0.f}, psA[4] = {0.f, 0.f, 0.f, 0.f}, xsB[2] = {0.f, 0.f}, psB[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned pwA[16], pwB[16];
  bf16x8_t kr[4];
  s16x4_t vr[4][2];
  {
    // S(lo) for both q blocks
#define S7_PART 9
#include "qp_attn_s7_iter.inc"
#undef S7_PART
    S7_FENCE_S(s0);
    const bool pre_n = ti_lo < ntp;
    const int t0_n = (pre_n ? ti_lo : ti_lo - ntp) * kKV, seg_n = pre_n ? P : n;
    if ((t0_n + kKV > seg_n) || (!pre_n && t0_n + kKV - 1 > q0wA)) {
      const int qlim_nA = pre_n ? 0x7fffffff : p.q_row0 + qiA, qlim_nB = pre_n ? 0x7fffffff : p.q_row0 + qiB;
      S7_MASK1(s0A, qlim_nA)
      S7_MASK1(s0B, qlim_nB)
    }
    mxA = s0A[0][0]; mxB = s0B[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { mxA = fmaxf(mxA, s0A[kb][r]); mxB = fmaxf(mxB, s0B[kb][r]); }
  }
  __builtin_amdgcn_s_barrier();          // every wave has read K(lo): step lo may overwrite that buffer with K(lo+2)

  auto rescale_A = [&](float alpha) {
#define S7_PART 5
#include "qp_attn_s7_iter.inc"
#undef S7_PART
  };
  auto rescale_B = [&](float alpha) {
#define S7_PART 6
#include "qp_attn_s7_iter.inc"
#undef S7_PART
  };
  // running-max update with the deferred rescale (per q block; per-lane test, cross-half max only in the rare branch)
#define S7_UPDATE1(ST)                                                                  \
  if (!__all((mx##ST - m_run##ST) * c <= 8.0f)) {                                       \
    mx##ST = xhalf_max(mx##ST);                                                         \
    const float m_new = fmaxf(m_run##ST, mx##ST);                                       \
    const float alpha = __builtin_amdgcn_exp2f((m_run##ST - m_new) * c);                \
    m_run##ST = m_new;                                                                  \
    rs##ST *= alpha;                                                                    \
    S7_FENCE_O();                                                                       \
    rescale_##ST(alpha);                                                                \
    nmc##ST = -(m_run##ST * c);                                                         \
  }
  nmcA = -(m_runA * c); nmcB = -(m_runB * c);
  S7_UPDATE1(A)
  S7_UPDATE1(B)

#define S7_STEP_SCALARS(T)                                                              \
  const int tn_ = (T) + 1;                                                              \
  const bool pre_n = tn_ < ntp;                                                         \
  const int t0_n = (pre_n ? tn_ : tn_ - ntp) * kKV, seg_n = pre_n ? P : n;              \
  const int qlim_nA = pre_n ? 0x7fffffff : p.q_row0 + qiA, qlim_nB = pre_n ? 0x7fffffff : p.q_row0 + qiB; \
  const bool mask_n = (t0_n + kKV > seg_n) || (!pre_n && t0_n + kKV - 1 > q0wA);        \
  const __amdgpu_buffer_rsrc_t rk2 = k_rsrc((T) + 2), rv1 = v_rsrc((T) + 1);            \
  const int soff_k2 = tile_soff((T) + 2), soff_v1 = tile_soff((T) + 1);                 \
  const int vdma_off = kVB + (1 - vi) * 16384;                                          \
  const unsigned vrd = voffv + kVB + vi * 16384;

#ifdef QP_S7_TIMING
  long long tacc[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define S7_STAMP(K) { const long long t_ = __builtin_readcyclecounter(); tacc[K] += t_ - tlast; tlast = t_; }
#else
#define S7_STAMP(K)
#endif
#define S7_STEP_END()                                                                   \
  S7_STAMP(2);                                                                          \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
  S7_STAMP(3);                                                                          \
  __builtin_amdgcn_s_barrier();                                                         \
  S7_STAMP(4);                                                                          \
  vi = 1 - vi;

  int vi = 0;
  for (int t = ti_lo; t < ti_hi; t += 2) {
    {
      S7_STEP_SCALARS(t)
#define S7_PART 0
#include "qp_attn_s7_iter.inc"
#undef S7_PART
      S7_STAMP(0);
#define S7_PART 2
#include "qp_attn_s7_iter.inc"
#undef S7_PART
      S7_STAMP(1);
      if (t + 1 < ti_hi) { S7_UPDATE1(A) S7_UPDATE1(B) }
      S7_STEP_END()
    }
    if (t + 1 >= ti_hi) break;
    {
      S7_STEP_SCALARS(t + 1)
#define S7_PART 1
#include "qp_attn_s7_iter.inc"
#undef S7_PART
      S7_STAMP(0);
#define S7_PART 3
#include "qp_attn_s7_iter.inc"
#undef S7_PART
      S7_STAMP(1);
      if (t + 2 < ti_hi) { S7_UPDATE1(A) S7_UPDATE1(B) }
      S7_STEP_END()
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup
  S7_FENCE_O();
#ifdef QP_S7_TIMING
  if (blockIdx.x == 8 && lane == 0)
    printf("wave %d tiles %d: Q part %lld  P part %lld  update %lld  dma-wait %lld  barrier %lld  (clk per step)\n", wave, ti_hi - ti_lo,
           tacc[0] / (ti_hi - ti_lo), tacc[1] / (ti_hi - ti_lo), tacc[2] / (ti_hi - ti_lo), tacc[3] / (ti_hi - ti_lo), tacc[4] / (ti_hi - ti_lo));
#endif

  // epilogue: q block A is 32-row block 2*wave of the item, B is block 2*wave+1 (same partial layout as the 8-wave s6 form)
#define S7_STORE(ST, BLK, QI)                                                           \
  {                                                                                     \
    f32x16_t o[4];                                                                      \
    readout_##ST(o);                                                                    \
    const float l_run = xhalf_sum(rs##ST);                                              \
    if (partial) {                                                                      \
      f32x4_t* wo = reinterpret_cast<f32x4_t*>(w_part) + ((BLK) * 16) * 64 + lane;      \
      _Pragma("unroll") for (int db = 0; db < 4; ++db)                                   \
        _Pragma("unroll") for (int r4 = 0; r4 < 4; ++r4)                                 \
          wo[(db * 4 + r4) * 64] = (f32x4_t){o[db][r4 * 4 + 0], o[db][r4 * 4 + 1], o[db][r4 * 4 + 2], o[db][r4 * 4 + 3]}; \
      float* wm = w_part + kRows7 * 128 + (BLK) * 128 + lane;                           \
      wm[0] = m_run##ST; wm[64] = l_run;                                                \
    } else if ((QI) < nq) {                                                             \
      const float inv = 1.0f / l_run;                                                   \
      uint2* op = p.out + ((int64_t)(QI) * p.hq + head) * 32;                           \
      _Pragma("unroll") for (int db = 0; db < 4; ++db)                                   \
        _Pragma("unroll") for (int r4 = 0; r4 < 4; ++r4) {                               \
          bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv),            \
                        (__bf16)(o[db][r4 * 4 + 2] * inv), (__bf16)(o[db][r4 * 4 + 3] * inv)};            \
          op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);                      \
        }                                                                               \
    }                                                                                   \
  }
  auto readout_A = [&](f32x16_t* o) {
#define S7_PART 7
#include "qp_attn_s7_iter.inc"
#undef S7_PART
  };
  auto readout_B = [&](f32x16_t* o) {
#define S7_PART 8
#include "qp_attn_s7_iter.inc"
#undef S7_PART
  };
  float* w_part = partial ? p.ws + ((int64_t)(kvh * (p.items - p.n_whole) + (item - p.n_whole)) * p.nsplit + split) * partial_floats(kRows7)
                          : nullptr;
  S7_STORE(A, 2 * wave, qiA)
  S7_STORE(B, 2 * wave + 1, qiB)
}

}  // namespace

void qp_launch_attn_s7(const AttnParams& p, bool xcd, unsigned per_kvh, hipStream_t s) {
  const int G = xcd ? 8 / p.hkv : 1;
  const dim3 grid = xcd ? dim3(8 * ((per_kvh + G - 1) / G)) : dim3(per_kvh, (unsigned)p.hkv);
  if (xcd) attn_fwd_kernel_s7<true><<<grid, 256, 0, s>>>(p);
  else attn_fwd_kernel_s7<false><<<grid, 256, 0, s>>>(p);
}
