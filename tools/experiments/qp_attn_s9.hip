// Seam 3, "s9": attn_fwd_kernel_s9 with the softmax one tile further ahead of the P.V product that consumes it (a second packed-P
// buffer), so that BOTH halves of a tile step carry one score element per MFMA gap — s6 has to finish P(t) before its P part and piles
// the VALU work into the Q part (1355-1775 clk for 1024 clk of MFMA).  Same math, operand layouts, LDS images and DMA scheme as s6;
// schedule and register roles are described in tools/gen_attn_s9.py.  Differences in the arithmetic ORDER only: when the lazy rescale
// switches the reference, the O accumulators follow one tile later (alpha_pend), after the last old-reference P has been added.
#include "qp_attn.h"
#include <cstdlib>

using namespace qpattn;

namespace {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#define PIN() __builtin_amdgcn_sched_barrier(0)
// opaque use+def: the value must exist HERE (LLVM otherwise sinks the softmax arithmetic past the mask branch to its first use)
#define KEEP(X) asm volatile("" : "+v"(X))
#define KEEP2(X, Y) asm volatile("" : "+v"(X), "+v"(Y))
#define DSR_B128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DSR_TR16(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define WAIT_LGKM1(N, R0) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(R0) : "n"(N))
#define WAIT_LGKM2(N, R0, R1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(R0), "+v"(R1) : "n"(N))

// ---- the pieces the generated iteration is made of -----------------------------------------------------------------------
#define S9_KREAD(SLOT, KK, IMM) DSR_B128(kr[SLOT], koffv[KK], IMM)
// counted wait for the hand-issued LDS reads.  No register operands on purpose: an asm that "defines" the fragment makes hipcc
// put a hazard s_nop before the MFMA that reads it; the order wait -> MFMA is held by the sched_barrier between them.
#define S9_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory")
#define S9_QK(SLOT, KK, ACC) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr[SLOT], qf[KK], ACC, 0, 0, 0)
#define S9_VREAD(SLOT, ADDR, O0, O1) { DSR_TR16(vr[SLOT][0], ADDR, O0); DSR_TR16(vr[SLOT][1], ADDR, O1); }
#define S9_PV(SLOT, C, DB, PW)                                                                                       \
  {                                                                                                                  \
    const s16x4_t v0_ = vr[SLOT][0], v1_ = vr[SLOT][1];                                                              \
    const s16x8_t av_ = {v0_[0], v0_[1], v0_[2], v0_[3], v1_[0], v1_[1], v1_[2], v1_[3]};                            \
    const u32x4v pc_ = {PW[4 * (C)], PW[4 * (C) + 1], PW[4 * (C) + 2], PW[4 * (C) + 3]};                             \
    o[DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av_), __builtin_bit_cast(bf16x8_t, pc_), \
                                                    o[DB], 0, 0, 0);                                                 \
  }
// softmax element pipeline (stages of different elements share a gap; see tools/gen_attn_s6.py):
//   A: x = s*c - m*c      B: p = 2^x      C: row sum += p      PACK: two probabilities -> one bf16x2 word of the P.V operand
#define S9_ELA(S, E, XI) xs[XI] = __builtin_fmaf(S[(E) >> 4][(E) & 15], c, nmc)
#define S9_ELB(XI, PI) ps[PI] = __builtin_amdgcn_exp2f(xs[XI])
#define S9_ELC(PI) rs += ps[PI]
#define S9_PACK(PW, W, PA, PB) { PW[W] = pack_bf16(ps[PA], ps[PB]); KEEP(PW[W]); }
#define S9_ELKEEP() asm volatile("" : "+v"(xs[0]), "+v"(xs[1]), "+v"(ps[0]), "+v"(ps[1]), "+v"(ps[2]), "+v"(ps[3]), "+v"(rs))
// K/V tile rows straight into LDS (LDS-DMA: `buffer_load_dwordx4 ... lds`, 1 KB = 4 rows per wave instruction, destination
// M0 + lane*16, so the K slot swizzle / the V [key/4][d/32][key%4][32] image are applied on the SOURCE address of each lane).
// No staging registers, no ds_write traffic; completion is on vmcnt.
#define S9_DMA(RSRC, LDSOFF, VOFF, SOFF) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RSRC, (__attribute__((address_space(3))) void*)(lds3 + (LDSOFF)), 16, VOFF, SOFF, 0, 0)
#define S9_DMA_K(K, LDSBASE) \
  if constexpr ((K) < kPieces) S9_DMA(rk3, (LDSBASE) + (K) * (kWaves * 1024) + wave * 1024, ksrc_off + (K) * (kWaves * 1024), soff_k3)
#define S9_DMA_V(K) \
  if constexpr ((K) < kPieces) S9_DMA(rv1, vdma_off + (K) * (kWaves * 1024) + wave * 1024, vsrc_off + (K) * (kWaves * 1024), soff_v1)
// row max of the next tile's scores, 8 elements per call
#define S9_MAX4(S, Q)                                                                   \
  {                                                                                     \
    if ((Q) == 0) mx = S[0][0];                                                         \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) mx = fmaxf(mx, S[(Q) >> 1][((Q) & 1) * 8 + e_]); \
    KEEP(mx);                                                                           \
  }
#define S9_MASK_NEXT(S)                                                                 \
  if (mask_n) {                                                                         \
    _Pragma("unroll") for (int kb_ = 0; kb_ < 2; ++kb_)                                  \
      _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) {                                \
        const int jk_ = t0_n + kb_ * 32 + (r_ & 3) + 8 * (r_ >> 2) + 4 * hi;             \
        S[kb_][r_] = (jk_ < seg_n && jk_ <= qlim_n) ? S[kb_][r_] : -INFINITY;            \
      }                                                                                 \
  }

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

constexpr int kKB0 = 0, kKB1 = 16384, kVB = 32768;   // LDS: K0 | K1 | V0 | V1 [| V2]  (16 KB each; K offsets are immediates in tools/gen_attn_s6.py)

template <bool kXcd, int kWaves>
__global__ __launch_bounds__(64 * kWaves, kWaves == 4 ? 2 : 1) void attn_fwd_kernel_s9(AttnParams p) {
  constexpr int kRows = 32 * kWaves;                     // query rows per workgroup
  constexpr int kPieces = 16 / kWaves;                   // 1-KB LDS-DMA pieces per wave, tile and matrix
  constexpr int kNVB = 2;                                // V tile buffers
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kVB + kNVB * 16384];
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
  const int q0l = qb * kRows + wave * 32;
  const int qi = q0l + (lane & 31);
  const int q0w = p.q_row0 + q0l;
  const int hi = lane >> 5, l31 = lane & 31;
  const int n = (int)p.n, P = (int)p.P, nq = p.nq;

  int blk_end = qb * kRows + kRows;
  if (blk_end > nq) blk_end = nq;
  blk_end += p.q_row0;
  const int ntp = (P + kKV - 1) / kKV, ntt = (blk_end + kKV - 1) / kKV, nt = ntp + ntt;
  int ti_lo = 0, ti_hi = nt;
  if (partial) { ti_lo = (int)((int64_t)split * nt / p.nsplit); ti_hi = (int)((int64_t)(split + 1) * nt / p.nsplit); }
  // Tiles THIS wave computes: [ti_lo, ti_end).  Past its last visible key tile (causal diagonal of its 32 rows; at once for a wave whose
  // rows all lie past the last query of a ragged final block) every probability is 0, so the wave only keeps the DMA / barrier protocol of
  // the workgroup going (loop at the end) instead of burning MFMA energy on them: the kernel is power-bound.  Results are
  // bit-identical: those tiles contributed exact zeros and left the running max alone.
  int ti_end = ti_hi;
  if (p.prio_mode & 12) {                  // QP_S9_EARLY_OUT=0 walks every tile (A/B): bit 2 = rows past the last query, bit 3 = causal diagonal
    const int own_idle = (p.prio_mode & 4) && q0l >= nq ? ti_lo : ti_hi;
    const int own_causal = (p.prio_mode & 8) ? ntp + (q0w + 31) / kKV + 1 : ti_hi;
    const int own = own_idle < own_causal ? own_idle : own_causal;
    ti_end = own < ti_hi ? (own > ti_lo ? own : ti_lo) : ti_hi;
  }
  const uint4* kp_base = p.kp + (int64_t)kvh * p.pre_hs16;
  const uint4* vp_base = p.vp + (int64_t)kvh * p.pre_hs16;
  const uint4* kn_base = p.kn + (int64_t)kvh * p.new_hs16;
  const uint4* vn_base = p.vn + (int64_t)kvh * p.new_hs16;
  // K/V rows of tile tj: descriptor of its segment (rows past the segment end read as 0) + scalar byte offset of the tile
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

  bf16x8_t qf[8];
  {
    const int qrow = qi < nq ? qi : nq - 1;
    const uint4* qp = p.q + ((int64_t)qrow * p.hq + head) * 16;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = __builtin_bit_cast(bf16x8_t, qp[kk * 2 + hi]);
  }
  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) o[db] = (f32x16_t){0};
  float m_run = -1e30f;
  const float c = p.c;

  // ---- prologue: K(lo) -> K0, K(lo+1) -> K1, V(lo) -> V0 ---------------------------------------------------------------------
  auto dma_tile = [&](const __amdgpu_buffer_rsrc_t& rs_, int ldsbase, int srcoff, int soff) {
#pragma unroll
    for (int it = 0; it < kPieces; ++it) S9_DMA(rs_, ldsbase + it * (kWaves * 1024) + wave * 1024, srcoff + it * (kWaves * 1024), soff);
  };
  dma_tile(k_rsrc(ti_lo), kKB0, ksrc_off, tile_soff(ti_lo));
  dma_tile(v_rsrc(ti_lo), kVB, vsrc_off, tile_soff(ti_lo));
  dma_tile(k_rsrc(ti_lo + 1), kKB1, ksrc_off, tile_soff(ti_lo + 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x16_t s0[2], s1[2];
  float rs = 0.f, mx, nmc, xs[2] = {0.f, 0.f}, ps[4] = {0.f, 0.f, 0.f, 0.f};
  float alpha_pend = 1.f;
  bool pend = false;                                     // wave-uniform: O still has to follow the last reference switch
  unsigned pw0[16], pw1[16];
  bf16x8_t kr[4];
  s16x4_t vr[4][2];
  // scores of tile tj from K buffer `kb` into S (plain loop: prologue only), masked like S9_MASK_NEXT; tiles at or past ti_hi (another
  // split's, or none) are masked out entirely
  auto plain_scores = [&](f32x16_t (&S)[2], int kb, int tj) {
    S[0] = (f32x16_t){0}; S[1] = (f32x16_t){0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      bf16x8_t a0, a1;
      if (kb == 0) { DSR_B128(a0, koffv[kk], kKB0); DSR_B128(a1, koffv[kk], kKB0 + 8192); }
      else { DSR_B128(a0, koffv[kk], kKB1); DSR_B128(a1, koffv[kk], kKB1 + 8192); }
      WAIT_LGKM2(0, a0, a1);
      S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[kk], S[0], 0, 0, 0);
      S[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[kk], S[1], 0, 0, 0);
    }
    const bool pre = tj < ntp;
    const int t0 = (pre ? tj : tj - ntp) * kKV, seg = tj >= ti_hi ? 0 : (pre ? P : n);
    if ((t0 + kKV > seg) || (!pre && t0 + kKV - 1 > q0w)) {
      const int qlim = pre ? 0x7fffffff : p.q_row0 + qi;
#pragma unroll
      for (int kb_ = 0; kb_ < 2; ++kb_)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jk = t0 + kb_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          S[kb_][r] = (jk < seg && jk <= qlim) ? S[kb_][r] : -INFINITY;
        }
    }
    mx = S[0][0];
#pragma unroll
    for (int kb_ = 0; kb_ < 2; ++kb_)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[kb_][r]);
  };
  // Lazy rescale (as s4/s6): keep the old reference while the next tile's max stays within 2^8 of it for every row of the wave.  The
  // switch is decided BEFORE the softmax of that tile starts; the row sums switch at once, the O accumulators one tile later, after
  // the last P computed under the old reference has been added (alpha_pend).
#define S9_APPLY_PEND()                                                                 \
  if (pend) {                                                                           \
    _Pragma("unroll") for (int db = 0; db < 4; ++db)                                     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) o[db][r] *= alpha_pend;            \
    pend = false;                                                                       \
  }
#define S9_UPDATE_MAX()                                                                 \
  if (!__all((mx - m_run) * c <= 8.0f)) {                                               \
    mx = xhalf_max(mx);                                                                 \
    const float m_new = fmaxf(m_run, mx);                                               \
    alpha_pend = __builtin_amdgcn_exp2f((m_run - m_new) * c);                           \
    m_run = m_new;                                                                      \
    rs *= alpha_pend;                                                                   \
    pend = true;                                                                        \
    nmc = -(m_run * c);                                                                 \
  }
  plain_scores(s0, 0, ti_lo);                            // S(lo)
  nmc = -(m_run * c);
  S9_UPDATE_MAX();                                       // first reference (O and the row sums are still zero)
  pend = false;
  __builtin_amdgcn_s_barrier();                          // every wave has read K(lo): K0 may take K(lo+2)
  dma_tile(k_rsrc(ti_lo + 2), kKB0, ksrc_off, tile_soff(ti_lo + 2));
  // pre-step: P(lo) -> pw0 (plain code, once per item), S(lo+1) -> s1 and its row max
#pragma unroll
  for (int e = 0; e < 32; e += 2) {
    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e >> 4][e & 15], c, nmc));
    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[(e + 1) >> 4][(e + 1) & 15], c, nmc));
    rs += p0; rs += p1;
    pw0[e >> 1] = pack_bf16(p0, p1);
  }
  plain_scores(s1, 1, ti_lo + 1);                        // S(lo+1)
  S9_UPDATE_MAX();                                       // reference for softmax(lo+1); O follows after P(lo) has been added
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // K(lo+2) landed, every wave has read K(lo+1)

#define S9_STEP_SCALARS(T)                                                              \
  const int tn_ = (T) + 2;                                                              \
  const bool pre_n = tn_ < ntp;                                                         \
  const int t0_n = (pre_n ? tn_ : tn_ - ntp) * kKV, seg_n = tn_ >= ti_hi ? 0 : (pre_n ? P : n); \
  const int qlim_n = pre_n ? 0x7fffffff : p.q_row0 + qi;                                \
  const bool mask_n = (t0_n + kKV > seg_n) || (!pre_n && t0_n + kKV - 1 > q0w);         \
  const __amdgpu_buffer_rsrc_t rk3 = k_rsrc((T) + 3), rv1 = v_rsrc((T) + 1);            \
  const int soff_k3 = tile_soff((T) + 3), soff_v1 = tile_soff((T) + 1);                 \
  const int vi1 = vi ^ 1;                                                               \
  const int vdma_off = kVB + vi1 * 16384;                                               \
  const unsigned vrd_pref = voffv + kVB + vi * 16384, vrd_main = vrd_pref;              /* V(t) */

#define S9_STEP_END()                                                                   \
  {                                                                                     \
    S9_APPLY_PEND();                                                                    \
    S9_UPDATE_MAX();                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          /* own DMA pieces landed */ \
    S9_STAMP(3);                                                                        \
    __builtin_amdgcn_s_barrier();                                                       \
    S9_STAMP(0);                                                                        \
  }

#ifdef QP_S9_TIMING
  long long tacc[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define S9_STAMP(K) { const long long t_ = __builtin_readcyclecounter(); tacc[K] += t_ - tlast; tlast = t_; }
#else
#define S9_STAMP(K)
#endif
  if (kWaves == 8 && (p.prio_mode & 3) == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (kWaves == 8 && (p.prio_mode & 3) == 2 && wave < 4) __builtin_amdgcn_s_setprio(1);
  int vi = 0;                                              // V buffer that holds tile t
  for (int t = ti_lo; t < ti_end; t += 2) {
    {
      S9_STEP_SCALARS(t)
#define S9_PART 0
#include "qp_attn_s9_iter.inc"
#undef S9_PART
#define S9_PART 2
#include "qp_attn_s9_iter.inc"
#undef S9_PART
      S9_STAMP(2);
      S9_STEP_END()
      vi = vi1;
    }
    if (t + 1 >= ti_end) break;
    {
      S9_STEP_SCALARS(t + 1)
#define S9_PART 1
#include "qp_attn_s9_iter.inc"
#undef S9_PART
#define S9_PART 3
#include "qp_attn_s9_iter.inc"
#undef S9_PART
      S9_STAMP(2);
      S9_STEP_END()
      vi = vi1;
    }
  }
  S9_APPLY_PEND();
  for (int t = ti_end; t < ti_hi; ++t) {                   // steps this wave has no keys in: its share of the tile DMA + the step barrier
    const int vi1 = vi ^ 1;
    dma_tile(k_rsrc(t + 3), ((t - ti_lo) & 1) ? kKB0 : kKB1, ksrc_off, tile_soff(t + 3));
    dma_tile(v_rsrc(t + 1), kVB + vi1 * 16384, vsrc_off, tile_soff(t + 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vi = vi1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup

#ifdef QP_S9_TIMING
  if (blockIdx.x == 8 && lane == 0)
    printf("wave %d tiles %d: barrier-wait %lld  phaseQ %lld  phaseP %lld  tail %lld  (clk per tile)\n", wave, ti_hi - ti_lo,
           tacc[0] / (ti_hi - ti_lo), tacc[1] / (ti_hi - ti_lo), tacc[2] / (ti_hi - ti_lo), tacc[3] / (ti_hi - ti_lo));
#endif
  const float l_run = xhalf_sum(rs);
  if (partial) {
    float* w = p.ws + ((int64_t)(kvh * (p.items - p.n_whole) + (item - p.n_whole)) * p.nsplit + split) * partial_floats(kRows);
    f32x4_t* wo = reinterpret_cast<f32x4_t*>(w) + (wave * 16) * 64 + lane;   // [wave][reg/4][lane][4], then (m, l) [wave][2][lane]
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) wo[(db * 4 + r4) * 64] = (f32x4_t){o[db][r4 * 4 + 0], o[db][r4 * 4 + 1], o[db][r4 * 4 + 2], o[db][r4 * 4 + 3]};
    float* wm = w + kRows * 128 + wave * 128 + lane;
    wm[0] = m_run; wm[64] = l_run;
    return;
  }
  if (qi < nq) {
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

void qp_launch_attn_s9(const AttnParams& p, bool xcd, unsigned per_kvh, hipStream_t s) {
  const int G = xcd ? 8 / p.hkv : 1;
  const dim3 grid = xcd ? dim3(8 * ((per_kvh + G - 1) / G)) : dim3(per_kvh, (unsigned)p.hkv);
  if (p.qb_rows == 256) {
    if (xcd) attn_fwd_kernel_s9<true, 8><<<grid, 512, 0, s>>>(p);
    else attn_fwd_kernel_s9<false, 8><<<grid, 512, 0, s>>>(p);
  } else {
    if (xcd) attn_fwd_kernel_s9<true, 4><<<grid, 256, 0, s>>>(p);
    else attn_fwd_kernel_s9<false, 4><<<grid, 256, 0, s>>>(p);
  }
}
