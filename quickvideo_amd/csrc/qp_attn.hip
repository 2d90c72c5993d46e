// Seam 3: prefill attention of one group over (pruned prefix, new tokens) — MFMA, gfx950.
// Reference: qwen25_lvu.py:61-62 (repeat_kv) + :102-112 (flash_attn causal, bottom-right aligned).
//
// Structure (wave64, v_mfma_f32_32x32x16_bf16; layouts verified on hardware by tools/probe/probe_layouts.hip):
//   workgroup = 4 waves = 128 query rows of ONE q head; each wave owns 32 query rows; 165 VGPRs -> 3 workgroups per CU.
//   S^T = K.Q^T ("swapped" QK^T): A = K tile rows (keys), B = Q^T -> every lane holds 16 of the 32 keys of ONE query
//        (lane&31), so the softmax row reduction is in-lane + one permlane32_swap.
//   O^T = V^T.P : B = P straight from the S^T accumulator registers (the contraction order over keys is permuted
//        identically on both operands), A = V^T fetched with ds_read_b64_tr_b16 from a [key/4][d/32][key%4][32] LDS
//        image -> O^T accumulators keep query = lane, so the online-softmax rescale is lane-local too.
//   K tile in LDS row-major [64][128] with the 16-B slot index XOR (row&15): conflict-free ds_read_b128.
//   KV is walked as two segments in ONE loop with ONE instance of the tile body: prefix rows [0,P) (no causal mask)
//   then the group's new rows (causal).  K/V tiles come in with buffer loads (tile offset in the scalar soffset,
//   rows past the segment end read as zero).  Work items are (kv head, q block, kv split): a 1-D grid maps workgroup
//   b (XCD b%8) to a kv head so that an XCD's private L2 serves one kv head; the heaviest q blocks run first; items
//   that would form a ragged last round are split along KV (partials in a caller-owned workspace + combine kernel).
#include "qp_attn.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

using namespace qpattn;

namespace {

// ------------------------------------------------------------------------------------------------
// Production kernel.  kXcd: 1-D grid with the XCD/kv-head mapping (needs 8 % Hkv == 0); otherwise grid = (items, Hkv).
// ------------------------------------------------------------------------------------------------
template <bool kXcd, int D, bool kVit>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_s4(AttnParams p) {
  constexpr int KSTEPS = D / 16;                          // 16-wide contraction steps of QK^T
  constexpr int NDB = (D + 31) / 32;                      // 32-wide d blocks of O^T
  constexpr int SLOTS = D / 8;                            // 16-B slots per K/V row
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kKV * kD * 2];
  unsigned char* kl = lds;
  unsigned char* vl = lds + kKV * kD * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int kvh, j;                                             // j = workgroup index inside its kv head
  if (kXcd) {
    // Workgroup b is observed to run on XCD b % 8 (used for speed only, never for correctness).  G = 8/Hkv XCDs serve one
    // kv head: all workgroups resident on an XCD stream the same K/V rows, so they hit in that XCD's private L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (kVit) {
      // every q block of one (sequence, head) pair runs on the same XCD: its K/V rows are fetched into one L2, not eight
      kvh = (slot / p.items) * 8 + xcd;
      j = slot % p.items;
      if (kvh >= p.hkv) return;
    } else {
      const int G = 8 / p.hkv;
      kvh = xcd / G;
      j = slot * G + (xcd % G);
    }
  } else {
    kvh = blockIdx.y; j = blockIdx.x;
  }
  int item, split = 0;
  const bool partial = j >= p.n_whole;
  if (!partial) item = j;
  else { const int r = j - p.n_whole; item = p.n_whole + r / p.nsplit; split = r % p.nsplit; }
  if (item >= p.items) return;
  const int qb = p.nqb - 1 - item / p.group;               // heaviest (latest) q blocks first
  const int head = kvh * p.group + item % p.group;
  const int q0l = qb * kQB + wave * 32;                      // local query row of this wave (q / out indexing)
  const int qi = q0l + (lane & 31);
  const int q0w = p.q_row0 + q0l;                            // row inside the group's new segment (causal mask)
  const int hi = lane >> 5, l31 = lane & 31;
  const int n = (int)p.n;
  int P = (int)p.P, nq = p.nq;
  const int seq = kVit ? kvh / p.heads_per_seq : 0;
  int64_t seq_row0 = kVit ? (int64_t)seq * n : 0;            // first row of this sequence in the packed [rows][3][H][D] tensor
  if (kVit && p.cu_seqlens) {                              // ragged batch: this sequence's rows and length
    const int a = p.cu_seqlens[seq];
    seq_row0 = a;
    P = nq = p.cu_seqlens[seq + 1] - a;
    if (qb * kQB >= nq) return;                            // workgroup-uniform, before any barrier
  }

  int blk_end = qb * kQB + kQB;
  if (blk_end > nq) blk_end = nq;
  blk_end += p.q_row0;                                        // new keys this workgroup can see: [0, blk_end)
  const int ntp = (P + kKV - 1) / kKV, ntt = kVit ? 0 : (blk_end + kKV - 1) / kKV, nt = ntp + ntt;
  int ti_lo = 0, ti_hi = nt;
  if (partial) { ti_lo = (int)((int64_t)split * nt / p.nsplit); ti_hi = (int)((int64_t)(split + 1) * nt / p.nsplit); }
  const int row_bytes = kVit ? p.kv_row_bytes : 256;
  const int64_t kv_base16 = kVit ? seq_row0 * (row_bytes / 16) + (int64_t)(kvh % p.heads_per_seq) * SLOTS : (int64_t)kvh * p.pre_hs16;
  const __amdgpu_buffer_rsrc_t rkp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kp + kv_base16), 0, P * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvp = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vp + kv_base16), 0, P * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rkn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.kn + (int64_t)kvh * p.new_hs16), 0, kVit ? 0 : n * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvn = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vn + (int64_t)kvh * p.new_hs16), 0, kVit ? 0 : n * 256, 0x00020000);

  const int r0 = tid >> 4, slot16 = tid & 15;            // this thread's rows r0 + 16*it, 16-B slot
  const int src_off = r0 * row_bytes + slot16 * 16;
  const int it_bytes = 16 * row_bytes;                    // rows advance by 16 per `it`
  const bool ld_on = slot16 < SLOTS;                      // D = 80: slots 10..15 of the 256-B LDS rows stay unused
  const int kdst = r0 * 256 + ((slot16 ^ (r0 & 15)) << 4);                                           // + it*4096
  const int vdst = (((r0 >> 2) * 4 + (slot16 >> 2)) << 8) + ((r0 & 3) << 6) + ((slot16 & 3) << 4);   // + it*4096
  int koff[8];                                            // LDS read addresses (tile independent)
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) koff[kk] = l31 * 256 + (((kk * 2 + hi) ^ (l31 & 15)) << 4);         // + kb*8192
  const int voff = (((lane & 15) >> 2) << 6) + (((lane >> 4) & 1) << 5) + ((lane & 3) << 3) + (hi << 10);

  bf16x8_t qf[KSTEPS];
  {
    const int qrow = qi < nq ? qi : nq - 1;
    const uint4* qp = kVit ? p.q + (seq_row0 + qrow) * (row_bytes / 16) + (int64_t)(kvh % p.heads_per_seq) * SLOTS
                           : p.q + ((int64_t)qrow * p.hq + head) * 16;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) qf[kk] = __builtin_bit_cast(bf16x8_t, qp[kk * 2 + hi]);
  }
  f32x16_t o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db) o[db] = (f32x16_t){0};
  float m_run = -1e30f, l_run = 0.f;
  const float c = p.c;

  for (int ti = ti_lo; ti < ti_hi; ++ti) {
    const bool pre = kVit || ti < ntp;
    const int t0 = (pre ? ti : ti - ntp) * kKV;
    const int seg_len = pre ? P : n;
    {
      const int soff = t0 * row_bytes;
      u32x4_t kv[4], vv[4];
      if (pre) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (SLOTS == 16 || ld_on) {
            kv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkp, src_off + it * it_bytes, soff, 0);
            vv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvp, src_off + it * it_bytes, soff, 0);
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          kv[it] = __builtin_amdgcn_raw_buffer_load_b128(rkn, src_off + it * 4096, soff, 0);
          vv[it] = __builtin_amdgcn_raw_buffer_load_b128(rvn, src_off + it * 4096, soff, 0);
        }
      }
      __syncthreads();                                  // everyone is done reading the previous tile
      if (SLOTS == 16 || ld_on) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          *reinterpret_cast<u32x4_t*>(kl + kdst + it * 4096) = kv[it];
          *reinterpret_cast<u32x4_t*>(vl + vdst + it * 4096) = vv[it];
        }
      }
      __syncthreads();
    }
    if (!pre && t0 > q0w + 31) continue;                // wave-uniform: no key of this tile is visible to this wave
    f32x16_t s[2];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      s[kb] = (f32x16_t){0};
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        bf16x8_t a = lds_read_b128(kl, koff[kk] + kb * 8192);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[kk], s[kb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    const bool need_mask = (t0 + kKV > seg_len) || (!pre && t0 + kKV - 1 > q0w);
    if (need_mask) {                                    // wave-uniform side branch: ragged or diagonal tiles only
      const int qlim = pre ? 0x7fffffff : p.q_row0 + qi;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jk = t0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          s[kb][r] = (jk < seg_len && jk <= qlim) ? s[kb][r] : -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = xhalf_max(mx);
    // deferred rescale: keep the old reference max while it is within 2^8 of the tile max for EVERY row of the wave
    if (!__all((mx - m_run) * c <= 8.0f)) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < NDB; ++db)
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
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const int off = voff + (((kb * 8 + cc * 4) * 4 + db) << 8);     // 4-key row group kq = kb*8 + cc*4 + hi
          s16x4_t v0 = lds_read_tr16(vl, off);
          s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));             // +8 keys = +2 row groups
          s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][cc], o[db], 0, 0, 0);
        }
      }
    __builtin_amdgcn_s_setprio(0);
  }
  if (partial) {
    // raw accumulators + (m, l) to the workspace; combined by attn_combine_kernel (same thread geometry)
    float* w = p.ws + ((int64_t)(kvh * (p.items - p.n_whole) + (item - p.n_whole)) * p.nsplit + split) * kPartialFloats;
    // layout [wave][reg/4][lane][4]: 16-B stores, 1 KB contiguous per wave instruction (combine reads the same way)
    f32x4_t* wo = reinterpret_cast<f32x4_t*>(w) + (wave * 16) * 64 + lane;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) wo[(db * 4 + r4) * 64] = (f32x4_t){o[db][r4 * 4 + 0], o[db][r4 * 4 + 1], o[db][r4 * 4 + 2], o[db][r4 * 4 + 3]};
    float* wm = w + 4 * 64 * 64 + wave * 128 + lane;
    wm[0] = m_run; wm[64] = l_run;
    return;
  }
  if (qi < nq) {
    const float inv = 1.0f / l_run;
    // LLM: out [nq][hq][128]; ViT: out [seq][S][heads][D]  (D/4 x 8 B per row)
    uint2* op = kVit ? p.out + ((seq_row0 + qi) * p.heads_per_seq + (kvh % p.heads_per_seq)) * (D / 4)
                     : p.out + ((int64_t)qi * p.hq + head) * 32;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        bf16x4_t v = {(__bf16)(o[db][r4 * 4 + 0] * inv), (__bf16)(o[db][r4 * 4 + 1] * inv), (__bf16)(o[db][r4 * 4 + 2] * inv),
                      (__bf16)(o[db][r4 * 4 + 3] * inv)};
        if (db * 32 + 8 * r4 + 4 * hi < D) op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);   // d = db*32 + 8*r4 + 4*hi
      }
  }
}

// merges the kv-split partials of one item: O = sum_s O_s 2^{(m_s-M)c} / sum_s l_s 2^{(m_s-M)c}.  One WAVE per (item, 32-row
// slice, 32-wide d block): grid (kv head x split items, slices, d blocks) of 64 threads, thread = one query x half as in the
// producer, 16-B loads; all (m, l) pairs are fetched first so the loads of every split are independent.  Slices past the last
// query row exit at once (a 30-token prompt tail has 28 split items: one workgroup per item took 65 us, this form 6).
constexpr int kMaxSplit = 16;
template <int D, bool kVit>
__global__ __launch_bounds__(64) void attn_combine_kernel(AttnParams p) {
  const int n_split_items = p.items - p.n_whole;
  const int kvh = blockIdx.x / n_split_items, it = blockIdx.x % n_split_items;
  const int item = p.n_whole + it;
  const int lane = threadIdx.x, wave = blockIdx.y, db = blockIdx.z, hi = lane >> 5;
  const int qb = p.nqb - 1 - item / p.group;
  const int head = kvh * p.group + item % p.group;
  if (qb * p.qb_rows + wave * 32 >= p.nq) return;
  const int qi = qb * p.qb_rows + wave * 32 + (lane & 31);
  const int pfl = partial_floats(p.qb_rows), o_floats = p.qb_rows * 128;
  const float* base = p.ws + (int64_t)(kvh * n_split_items + it) * p.nsplit * pfl;
  float ms[kMaxSplit], f[kMaxSplit];
  float M = -1e30f;
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) {
    ms[s] = -1e30f; f[s] = 0.f;
    if (s < p.nsplit) {
      const float* wm = base + (int64_t)s * pfl + o_floats + wave * 128 + lane;
      ms[s] = wm[0]; f[s] = wm[64];                      // f temporarily holds l_s
    }
  }
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) M = fmaxf(M, ms[s]);
  float L = 0.f;
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) {
    const float e = __builtin_amdgcn_exp2f((ms[s] - M) * p.c);
    L += f[s] * e;
    f[s] = e;
  }
  f32x4_t acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) {
    if (s < p.nsplit) {
      const f32x4_t* wo = reinterpret_cast<const f32x4_t*>(base + (int64_t)s * pfl) + (wave * 16 + db * 4) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += wo[r * 64] * f[s];
    }
  }
  if (qi < p.nq) {
    const float inv = 1.0f / L;
    uint2* op = kVit ? p.out + (((int64_t)(kvh / p.heads_per_seq) * p.n + qi) * p.heads_per_seq + (kvh % p.heads_per_seq)) * (D / 4)
                     : p.out + ((int64_t)qi * p.hq + head) * 32;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const f32x4_t a = acc[r4];
      bf16x4_t v = {(__bf16)(a[0] * inv), (__bf16)(a[1] * inv), (__bf16)(a[2] * inv), (__bf16)(a[3] * inv)};
      if (db * 32 + 8 * r4 + 4 * hi < D) op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
    }
  }
}

// Flat split (AttnParams::flat_pieces): merges the partial segments of one item.  An item whose tiles [F0, F0 + nt) lie in pieces
// j0..j1 has one part per piece; the part in piece j sits in slot 0 of that piece when the item is the piece's FIRST segment, else in
// slot 1 (a later segment that is partial is always the piece's last).  Items covered by a single whole-item segment were written by
// the producer and exit here.  Same thread geometry and accumulation as attn_combine_kernel.
__global__ __launch_bounds__(64) void attn_combine_flat_kernel(AttnParams p) {
  const int kvh = blockIdx.x / p.items, item = blockIdx.x % p.items;
  const int lane = threadIdx.x, wave = blockIdx.y, db = blockIdx.z, hi = lane >> 5;
  const int qb = p.nqb - 1 - item / p.group;
  const int head = kvh * p.group + item % p.group;
  if (qb * p.qb_rows + wave * 32 >= p.nq) return;
  const int nt = attn_qb_tiles(p, qb);
  const long long F0 = attn_flat_item_start(p, item), F1 = F0 + nt;
  auto piece_of = [&](long long f) {
    int j = (int)(f * p.flat_pieces / p.flat_total);
    while (j + 1 < p.flat_pieces && attn_flat_bound(p, j + 1) <= f) ++j;
    while (j > 0 && attn_flat_bound(p, j) > f) --j;
    return j;
  };
  const int j0 = piece_of(F0), j1 = piece_of(F1 - 1);
  if (j0 == j1 && attn_flat_bound(p, j0) <= F0 && attn_flat_bound(p, j0 + 1) >= F1) return;     // one whole-item segment: output already written
  const int parts = j1 - j0 + 1;                            // <= kMaxSplit (checked by the planner)
  const int qi = qb * p.qb_rows + wave * 32 + (lane & 31);
  const int pfl = partial_floats(p.qb_rows), o_floats = p.qb_rows * 128;
  int dummy;
  const int first_item_of_j0 = attn_flat_locate(p, attn_flat_bound(p, j0), &dummy);
  auto slot_base = [&](int s) {                             // part s lives in piece j0 + s
    const int slot = (s == 0 && first_item_of_j0 != item) ? 1 : 0;
    return p.ws + (((int64_t)kvh * p.flat_pieces + (j0 + s)) * 2 + slot) * pfl;
  };
  float ms[kMaxSplit], f[kMaxSplit];
  float M = -1e30f;
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) {
    ms[s] = -1e30f; f[s] = 0.f;
    if (s < parts) {
      const float* wm = slot_base(s) + o_floats + wave * 128 + lane;
      ms[s] = wm[0]; f[s] = wm[64];
    }
  }
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) M = fmaxf(M, ms[s]);
  float L = 0.f;
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) {
    const float e = __builtin_amdgcn_exp2f((ms[s] - M) * p.c);
    L += f[s] * e;
    f[s] = e;
  }
  f32x4_t acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < kMaxSplit; ++s) {
    if (s < parts) {
      const f32x4_t* wo = reinterpret_cast<const f32x4_t*>(slot_base(s)) + (wave * 16 + db * 4) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += wo[r * 64] * f[s];
    }
  }
  if (qi < p.nq) {
    const float inv = 1.0f / L;
    uint2* op = p.out + ((int64_t)qi * p.hq + head) * 32;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const f32x4_t a = acc[r4];
      bf16x4_t v = {(__bf16)(a[0] * inv), (__bf16)(a[1] * inv), (__bf16)(a[2] * inv), (__bf16)(a[3] * inv)};
      op[db * 8 + r4 * 2 + hi] = __builtin_bit_cast(uint2, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// v1: first correct kernel of round 1 (clamped global loads, 4 inlined tile bodies, always-rescale, plain 2-D grid).
// Kept as an independent implementation: QP_ATTN_VARIANT=1 cross-checks the production kernel in tools/bench_attn.py,
// and it is the fallback for segments >= 2 GiB per head (buffer descriptors address 32 bits).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_tile_v1(const uint4* __restrict__ ks, const uint4* __restrict__ vs, int64_t t0, int64_t seg_len,
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
__device__ __forceinline__ void tile_compute_v1(const unsigned char* kl, const unsigned char* vl, const bf16x8_t (&qf)[8],
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
  const int g1 = (lane >> 4) & 1;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int key0 = kb * 32 + cc * 16 + 4 * hi;
      const int kq = key0 >> 2;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const int off = ((kq * 4 + db) << 8) + (((lane & 15) >> 2) << 6) + (g1 << 5) + ((lane & 3) << 3);
        s16x4_t v0 = lds_read_tr16(vl, off);
        s16x4_t v1 = lds_read_tr16(vl, off + (2 * 4 << 8));
        s16x8_t av = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), pf[kb][cc], o[db], 0, 0, 0);
      }
    }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel_v1(AttnParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kKV * kD * 2];
  unsigned char* kl = lds;
  unsigned char* vl = lds + kKV * kD * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qb = gridDim.x - 1 - blockIdx.x;
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
  {
    const uint4* ks = p.kp + (int64_t)kvh * p.pre_hs16;
    const uint4* vs = p.vp + (int64_t)kvh * p.pre_hs16;
    for (int64_t t0 = 0; t0 < p.P; t0 += kKV) {
      __syncthreads();
      load_tile_v1(ks, vs, t0, p.P, kl, vl, tid);
      __syncthreads();
      if (t0 + kKV <= p.P) tile_compute_v1<false, false>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
      else tile_compute_v1<true, false>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.P, qi, lane);
    }
  }
  {
    const uint4* ks = p.kn + (int64_t)kvh * p.new_hs16;
    const uint4* vs = p.vn + (int64_t)kvh * p.new_hs16;
    int64_t blk_end = (int64_t)qb * kQB + kQB;
    if (blk_end > p.n) blk_end = p.n;
    for (int64_t t0 = 0; t0 < blk_end; t0 += kKV) {
      __syncthreads();
      load_tile_v1(ks, vs, t0, p.n, kl, vl, tid);
      __syncthreads();
      if (t0 <= q0w + 31) {
        if (t0 + kKV - 1 <= q0w && t0 + kKV <= p.n) tile_compute_v1<false, true>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
        else tile_compute_v1<true, true>(kl, vl, qf, o, m_run, l_run, p.c, t0, p.n, qi, lane);
      }
    }
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

// Work-item plan of one launch.  Per kv head there are `items` = nqb*group items (q block x q head), dispatched heaviest
// (latest q block) first onto S = CUs * workgroups-per-CU / Hkv resident workgroup slots.  The first `n_whole` items run
// unsplit; the ragged remainder is cut into `nsplit` kv ranges whose partials are merged by attn_combine_kernel.  n_whole and
// nsplit are chosen by simulating that dispatch (greedy: next workgroup to the first free slot) with per-item tile counts and
// taking the smallest makespan; the same estimate picks between the 4-wave (128-row, 2 per CU) and 8-wave (256-row, 1 per CU)
// forms of the s6 kernel, which differ mostly in how the grid quantises.  Plans are cached per shape.
struct AttnPlan { int items, n_whole, nsplit, rows; double cost; };

double simulate_makespan(int64_t nq, int64_t Peff, int group, int nqb, int rows, int slots, int n_whole, int ns, double c0) {
  const int items = nqb * group;
  std::vector<double> heap(slots, 0.0);                           // min-heap of slot finish times
  auto cmp = [](double x, double y) { return x > y; };
  double makespan = 0.0;
  auto place = [&](double w) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    double& t = heap.back();
    t += w;
    if (t > makespan) makespan = t;
    std::push_heap(heap.begin(), heap.end(), cmp);
  };
  const int64_t ntp = (Peff + kKV - 1) / kKV;
  auto tiles_of = [&](int item) {
    const int qb = nqb - 1 - item / group;
    int64_t end = (int64_t)(qb + 1) * rows; if (end > nq) end = nq;
    return (double)(ntp + (end + kKV - 1) / kKV);
  };
  for (int it = 0; it < n_whole; ++it) place(tiles_of(it) + c0);
  for (int it = n_whole; it < items; ++it) {
    const double t = tiles_of(it);
    for (int sp = 0; sp < ns; ++sp) place(t / ns + c0 + 1.0);     // + partial store
  }
  return makespan;
}

AttnPlan plan_items(int64_t n, int64_t P, int hq, int hkv, int cus, int split_mode, int wg_per_cu, int qb_rows, int force_split) {
  AttnPlan a;
  const int nqb = (int)((n + qb_rows - 1) / qb_rows), group = hq / hkv;
  a.items = nqb * group; a.rows = qb_rows;
  a.n_whole = a.items; a.nsplit = 1; a.cost = 0.0;
  const int slots = cus * wg_per_cu / hkv > 0 ? cus * wg_per_cu / hkv : 1;   // resident workgroups per kv head
  // prologue + epilogue of a workgroup, in tile steps.  8-wave form (one workgroup per CU: nothing overlaps its Q load / first DMA /
  // output store): 8 — with 6 the plan picked it for short pure-causal launches (n = 2880 or 2240, no prefix: 84 / 63 items) where
  // the 4-wave form measures 10-12 % faster (849 vs 773 TF, 714 vs 639 TF); long items are insensitive to it.
  const double c0 = wg_per_cu == 1 ? 8.0 : 4.0;
  const double tstep = qb_rows == 256 ? 0.95 : 1.0;              // measured: an 8-wave tile step is ~5 % shorter (half the DMA pieces per wave)
  a.cost = simulate_makespan(n, P, group, nqb, qb_rows, slots, a.items, 1, c0) * tstep;
  if (split_mode == 0) return a;
  {                                                                // experiment: EVERY item cut into ns kv ranges.  With 2 XCDs per kv head and
    const int ns = force_split;                                    // ns = 2, split s of every item runs on XCD parity s (item index = 2*slot +
    if (ns >= 2 && ns <= kMaxSplit) {                              // xcd%2), so each XCD streams only half of the K/V rows: HBM reads halve.
      a.n_whole = 0; a.nsplit = ns;
      a.cost = (simulate_makespan(n, P, group, nqb, qb_rows, slots, 0, ns, c0) + 3.0) * tstep;
      return a;
    }
  }
  const int64_t tiles_min = (P + kKV - 1) / kKV + 2;              // tiles of the lightest item (q block 0)
  int64_t cap = tiles_min / 4; if (cap < 1) cap = 1;              // keep >= 4 tiles per piece
  if (cap > kMaxSplit) cap = kMaxSplit;
  const int full = a.items / slots * slots;
  for (int pass = 0; pass < 2; ++pass) {
    const int nw = pass == 0 ? full : full - slots;               // split the last ragged round, or that and one full round
    if (nw < 0 || nw == a.items) continue;
    for (int ns = 2; ns <= (int)cap; ++ns) {
      const double c = (simulate_makespan(n, P, group, nqb, qb_rows, slots, nw, ns, c0) + 3.0) * tstep;   // + combine launch
      if (c < a.cost * 0.995) { a.cost = c; a.n_whole = nw; a.nsplit = ns; }
    }
  }
  return a;
}

// cached plans: same (shape, prefix) for every layer of a group
AttnPlan plan_cached(int64_t n, int64_t P, int hq, int hkv, int cus, int split_mode, int wg_per_cu, int qb_rows) {
  typedef std::tuple<int64_t, int64_t, int, int, int, int, int, int, int> Key;
  static std::map<Key, AttnPlan> cache;
  static std::mutex mu;
  const int force_split = qp_dev().attn_force_split.load(std::memory_order_relaxed);
  const Key k(n, P, hq, hkv, cus, split_mode, wg_per_cu, qb_rows, force_split);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(k);
  if (it != cache.end()) return it->second;
  if (cache.size() > 4096) cache.clear();
  const AttnPlan a = plan_items(n, P, hq, hkv, cus, split_mode, wg_per_cu, qb_rows, force_split);
  if (qp_dev().attn_debug.load(std::memory_order_relaxed))
    fprintf(stderr, "[qp_attn plan] n=%lld P=%lld hq=%d hkv=%d wg/cu=%d rows=%d: items=%d n_whole=%d nsplit=%d cost=%.1f\n", (long long)n,
            (long long)P, hq, hkv, wg_per_cu, qb_rows, a.items, a.n_whole, a.nsplit, a.cost);
  cache[k] = a;
  return a;
}

// Flat (stream-K) plan for one workgroup form: `pieces` = the resident workgroup slots of a kv head (ONE round, every slot the same
// number of tiles), cost in the units of plan_items (tile steps).  Not eligible (cost = inf) when a piece would be shorter than 16 tiles
// (prologue-dominated), an item would be spread over more than kMaxSplit pieces, or the flat positions would not fit 31 bits.
struct FlatPlan { int pieces; long long total; double cost; };
FlatPlan plan_flat(const AttnParams& base, int cus, int wg_per_cu, int qb_rows, bool forced) {
  AttnParams q = base;
  q.qb_rows = qb_rows; q.nqb = (int)((q.nq + qb_rows - 1) / qb_rows); q.items = q.nqb * q.group;
  FlatPlan f; f.pieces = 0; f.total = 0; f.cost = 1e300;
  const int slots = cus * wg_per_cu / q.hkv > 0 ? cus * wg_per_cu / q.hkv : 1;
  long long W = 0; int nt_max = 0;
  for (int qb = 0; qb < q.nqb; ++qb) { const int nt = attn_qb_tiles(q, qb); W += (long long)nt * q.group; if (nt > nt_max) nt_max = nt; }
  if (W <= 0 || W >= (1ll << 30)) return f;
  const double len = (double)W / slots;
  if (len < (forced ? 2.0 : 16.0) || nt_max / len + 2.0 > (double)kMaxSplit) return f;   // forced (tests, A/B): any range of >= 2 tiles
  if (!forced) {
    // Measured (tools/bench_attn_flat.py, profiles/r5_attn_flat_ab.txt): equal tile ranges start at a different key offset in every
    // workgroup, so the workgroups of an XCD no longer stream the SAME K/V tiles at the same time and the L2 sharing the item-granular
    // plan lives on is gone.  The flat form therefore only wins while all K/V rows of the launch sit in the Infinity Cache (<= 96 MB:
    // +4 % at n = 960..1024 over 20-50 k prefix rows; -3 % at 123 MB, -5 % at 216 k rows, -15 % on the 1-hour video's steady state) and
    // while a range holds at most two segments (items <= slots; with more, the per-segment prologues cost more than the balance gains:
    // -7 % at n = 2880).
    const double kv_bytes = (double)(q.P + q.n) * q.hkv * 2.0 * 256.0;
    if (kv_bytes > 96e6 || q.items > slots) return f;
  }
  const double c0 = wg_per_cu == 1 ? 8.0 : 4.0, tstep = qb_rows == 256 ? 0.95 : 1.0;
  const double segs = 1.0 + (double)q.items / slots;      // item boundaries inside a piece, on average, + 1
  f.pieces = slots; f.total = W;
  f.cost = (len + c0 * (segs < 2.0 ? 2.0 : segs + 1.0) + 1.0 + 3.0) * tstep;   // + partial stores, + combine launch
  return f;
}

}  // namespace

size_t qp_attn_workspace_bytes_impl(const qp_ctx* ctx, int64_t nq, int64_t prefix_len, int hq, int hkv) {
  size_t need = 0;
  for (int cfg = 0; cfg < 3; ++cfg) {                      // any of the kernels may serve the call: size for the largest plan
    const int wg = cfg == 0 ? 3 : cfg == 1 ? 2 : 1, rows = cfg == 2 ? 256 : 128;     // s4 | s6<4> | s6<8>
    AttnPlan a = plan_cached(nq, prefix_len, hq, hkv, ctx->cus, 1, wg, rows);
    const size_t b = (size_t)hkv * (size_t)(a.items - a.n_whole) * (size_t)a.nsplit * partial_floats(rows) * sizeof(float);
    if (b > need) need = b;
  }
  // the flat (stream-K) form — two partial slots per resident workgroup (69 MB on an unmasked MI355X) — only for the shapes the launcher
  // would take it for (by cost, or forced by the developer switch): every other shape keeps the size it had before round 5, and a launch
  // that finds less workspace than the flat form needs simply runs the item-granular plan
  const int flat_mode = qp_dev().attn_flat.load(std::memory_order_relaxed);
  if (flat_mode != 0 && hq % hkv == 0) {
    AttnParams q;
    q.P = prefix_len; q.n = nq; q.nq = (int)nq; q.q_row0 = 0; q.hq = hq; q.hkv = hkv; q.group = hq / hkv;
    double classic = 1e300;
    for (int cfg = 1; cfg < 3; ++cfg) {
      const AttnPlan a = plan_cached(nq, prefix_len, hq, hkv, ctx->cus, 1, cfg == 1 ? 2 : 1, cfg == 1 ? 128 : 256);
      if (a.cost < classic) classic = a.cost;
    }
    for (int form = 0; form < 2; ++form) {
      const int wg = form == 0 ? 2 : 1, rows = form == 0 ? 128 : 256;
      const FlatPlan f = plan_flat(q, ctx->cus, wg, rows, flat_mode == 1);
      if (f.pieces > 0 && (flat_mode == 1 || f.cost < 0.97 * classic)) {
        const size_t b = (size_t)hkv * (size_t)f.pieces * 2 * partial_floats(rows) * sizeof(float);
        if (b > need) need = b;
      }
    }
  }
  return need + 256;
}

int qp_launch_prefill_attn(const qp_ctx* ctx, const void* q, const void* k_prefix, const void* v_prefix,
                           int64_t prefix_head_stride, int64_t prefix_len, const void* k_new, const void* v_new,
                           int64_t new_head_stride, int64_t n, int64_t q_row0, int64_t nq, int hq, int hkv, float scale, void* out,
                           void* workspace, size_t workspace_bytes, hipStream_t s) {
  AttnParams p;
  p.flat_pieces = 0; p.flat_total = 0; p.variant = 0;
  p.q = (const uint4*)q; p.out = (uint2*)out;
  p.kp = (const uint4*)k_prefix; p.vp = (const uint4*)v_prefix; p.pre_hs16 = prefix_head_stride / 8; p.P = prefix_len;
  p.kn = (const uint4*)k_new; p.vn = (const uint4*)v_new; p.new_hs16 = new_head_stride / 8; p.n = n;
  p.hq = hq; p.group = hq / hkv; p.c = scale * 1.4426950408889634f;
  p.nqb = (int)((nq + kQB - 1) / kQB); p.hkv = hkv; p.ws = (float*)workspace;
  p.items = p.nqb * p.group; p.n_whole = p.items; p.nsplit = 1;
  p.heads_per_seq = hkv; p.seq_stride16 = 0; p.kv_row_bytes = 256; p.cu_seqlens = nullptr;
  const qp_dev_switches& dev = qp_dev();                         // developer A/B switches (qp_dev_switch; tools/bench_attn.py): atomics, no getenv here
  p.prio_mode = (dev.s6_prio.load(std::memory_order_relaxed) & 3) | ((dev.s6_early_out.load(std::memory_order_relaxed) & 3) << 2);   // see qp_attn_s6.hip
  p.q_row0 = (int)q_row0; p.nq = (int)nq; p.qb_rows = kQB;
  const int variant = dev.attn_variant.load(std::memory_order_relaxed);   // default 0 = production kernel
  p.variant = variant;
  const bool big = prefix_len * 256 >= (1ll << 31) || n * 256 >= (1ll << 31);
  if ((variant == 1 || big) && !(q_row0 == 0 && nq == n))
    return qp_fail(QP_ERR_UNSUPPORTED, "qp_prefill_attn: query sub-ranges need K/V segments below 2 GiB per head");
  if (variant == 1 || big) {
    attn_fwd_kernel_v1<<<dim3((unsigned)p.nqb, (unsigned)hq), 256, 0, s>>>(p);
    return qp_check_launch("prefill_attn(v1)");
  }
  // variant 2: no kv split; variant 3: no XCD mapping; variant 4: previous production kernel s4 (phase-sequential waves);
  // variant 7 / 8: s6 with 4-wave (128 rows, 2 per CU) / 8-wave (256 rows, 1 per CU) workgroups
  // default: whichever s6 form the dispatch simulation predicts to finish first; 8: force 8-wave
  const int split_mode = (variant == 2 || workspace == nullptr) ? 0 : 1;
  AttnPlan a;
  if (variant == 4) a = plan_cached(nq, prefix_len + q_row0, hq, hkv, ctx->cus, split_mode, 3, 128);
  else if (variant == 7) a = plan_cached(nq, prefix_len + q_row0, hq, hkv, ctx->cus, split_mode, 2, 128);
  else if (variant == 8 || variant == 9 || variant == 10) a = plan_cached(nq, prefix_len + q_row0, hq, hkv, ctx->cus, split_mode, 1, 256);
  else {
    a = plan_cached(nq, prefix_len + q_row0, hq, hkv, ctx->cus, split_mode, 2, 128);
    const AttnPlan b = plan_cached(nq, prefix_len + q_row0, hq, hkv, ctx->cus, split_mode, 1, 256);
    if (b.cost < a.cost) a = b;
  }
  // flat (stream-K) split: taken when one round of equal tile ranges beats the item-granular plan by > 3 % in the same cost model —
  // short groups over long prefixes (n <= 1024: 28-56 equal items for 64-128 slots, where an item-granular split leaves 1/8 of the
  // chip idle).  Developer switch attn_flat: 0 never, 1 whenever eligible, -1 (default) by cost.  s6 forms only; needs the workspace.
  p.flat_pieces = 0; p.flat_total = 0;
  const int flat_mode = dev.attn_flat.load(std::memory_order_relaxed);
  if (flat_mode != 0 && split_mode == 1 && variant != 4 && variant != 9 && variant != 10) {
    FlatPlan best; best.pieces = 0; best.cost = 1e300; int best_rows = 0;
    for (int form = 0; form < 2; ++form) {
      const int wg = form == 0 ? 2 : 1, rows_f = form == 0 ? 128 : 256;
      if ((variant == 7 && form == 1) || (variant == 8 && form == 0)) continue;
      const FlatPlan f = plan_flat(p, ctx->cus, wg, rows_f, flat_mode == 1);
      if (f.pieces > 0 && f.cost < best.cost) { best = f; best_rows = rows_f; }
    }
    const size_t need_f = best.pieces > 0 ? (size_t)hkv * best.pieces * 2 * partial_floats(best_rows) * sizeof(float) : 0;
    if (best.pieces > 0 && need_f <= workspace_bytes && (flat_mode == 1 || best.cost < 0.97 * a.cost)) {
      p.qb_rows = best_rows; p.nqb = (int)((nq + best_rows - 1) / best_rows);
      p.items = p.nqb * p.group; p.n_whole = p.items; p.nsplit = 1;
      p.flat_pieces = best.pieces; p.flat_total = best.total;
      const bool xcd_f = (hkv <= 8 && 8 % hkv == 0 && variant != 3);
      qp_launch_attn_s6(p, xcd_f, (unsigned)best.pieces, s);
      int rc_f = qp_check_launch("prefill_attn(flat)");
      if (rc_f) return rc_f;
      attn_combine_flat_kernel<<<dim3((unsigned)(hkv * p.items), (unsigned)(best_rows / 32), 4), 64, 0, s>>>(p);
      return qp_check_launch("prefill_attn(flat combine)");
    }
  }
  const int rows = a.rows;
  p.qb_rows = rows; p.nqb = (int)((nq + rows - 1) / rows);
  p.items = a.items; p.n_whole = a.n_whole; p.nsplit = a.nsplit;
  if (a.nsplit > 1) {
    const size_t need = (size_t)hkv * (size_t)(a.items - a.n_whole) * (size_t)a.nsplit * partial_floats(rows) * sizeof(float);
    if (workspace_bytes < need) return qp_fail(QP_ERR_WORKSPACE, "qp_prefill_attn: workspace %zu < %zu bytes", workspace_bytes, need);
  }
  const int per_kvh = a.n_whole + (a.items - a.n_whole) * a.nsplit;
  const bool xcd = (hkv <= 8 && 8 % hkv == 0 && variant != 3);
#ifdef QP_EXPERIMENTS                                    // `make EXPERIMENTS=1` (measured slower): not in the product library
  if (variant == 10) {                                   // experiment: one wave per SIMD, 64 rows per wave (qp_attn_s7.hip)
    qp_launch_attn_s7(p, xcd, (unsigned)per_kvh, s);
  } else
#else
  if (variant == 9 || variant == 10)
    return qp_fail(QP_ERR_UNSUPPORTED, "qp_prefill_attn: QP_ATTN_VARIANT=%d is an experiment; rebuild with `make EXPERIMENTS=1`", variant);
#endif
  if (variant != 4) {                             // production: software-pipelined kernel (qp_attn_s6.hip); 4: s4
    qp_launch_attn_s6(p, xcd, (unsigned)per_kvh, s);
  } else if (xcd) {
    const int G = 8 / hkv;
    attn_fwd_kernel_s4<true, 128, false><<<dim3(8 * ((per_kvh + G - 1) / G)), 256, 0, s>>>(p);
  } else {
    attn_fwd_kernel_s4<false, 128, false><<<dim3((unsigned)per_kvh, (unsigned)hkv), 256, 0, s>>>(p);
  }
  int rc = qp_check_launch("prefill_attn");
  if (rc) return rc;
  if (a.nsplit > 1) {
    attn_combine_kernel<128, false><<<dim3((unsigned)(hkv * (a.items - a.n_whole)), (unsigned)(rows / 32), 4), 64, 0, s>>>(p);
    rc = qp_check_launch("prefill_attn(combine)");
  }
  return rc;
}

// Batched non-causal attention of the ViT tower: qkv bf16 [n_seq*S][3][H][80] (after the rotary), out [n_seq*S][H][80].
// Every (sequence, head) pair is its own "kv head" with one q head; 1-D grid with all q blocks of a pair on one XCD (neutral
// at the video shapes, where the whole qkv fits the MALL), no kv split (n_seq*H*ceil(S/128) workgroups is several rounds).
// cu_seqlens != NULL: ragged batch (Qwen2.5-VL window attention) — n_seq sequences of at most S rows, sequence i = rows
// [cu[i], cu[i+1]) of the packed tensor; workgroups past a sequence's end exit at once.
int qp_launch_vit_attn(const qp_ctx* ctx, const void* qkv, int64_t n_seq, int64_t S, int heads, float scale, void* out,
                       const int* cu_seqlens, hipStream_t s) {
  (void)ctx;
  constexpr int D = 80;
  AttnParams p;
  p.flat_pieces = 0; p.flat_total = 0; p.variant = 0;
  const uint4* base = (const uint4*)qkv;
  const int row16 = 3 * heads * D / 8;
  p.q = base; p.kp = base + heads * D / 8; p.vp = base + 2 * heads * D / 8; p.kn = p.kp; p.vn = p.vp;
  p.out = (uint2*)out; p.pre_hs16 = 0; p.new_hs16 = 0; p.P = S; p.n = S;
  const int hk = (int)(n_seq * heads);
  p.hq = hk; p.group = 1; p.c = scale * 1.4426950408889634f;
  p.nqb = (int)((S + kQB - 1) / kQB); p.hkv = hk; p.ws = nullptr;
  p.heads_per_seq = heads; p.seq_stride16 = S * row16; p.kv_row_bytes = row16 * 16; p.cu_seqlens = cu_seqlens; p.prio_mode = 0;
  p.items = p.nqb; p.n_whole = p.nqb; p.nsplit = 1; p.q_row0 = 0; p.nq = (int)S; p.qb_rows = kQB;
  p.variant = qp_dev().attn_variant.load(std::memory_order_relaxed);      // 3: plain 2-D grid (A/B of the XCD mapping)
  if (p.variant == 3) attn_fwd_kernel_s4<false, D, true><<<dim3((unsigned)p.nqb, (unsigned)hk), 256, 0, s>>>(p);
  else attn_fwd_kernel_s4<true, D, true><<<dim3((unsigned)(((hk + 7) / 8) * 8 * p.nqb)), 256, 0, s>>>(p);
  return qp_check_launch("vit_attn");
}
