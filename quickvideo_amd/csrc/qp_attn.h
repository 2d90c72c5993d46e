// Internal definitions shared by the prefill-attention kernels (qp_attn.hip: production/ViT/v1 kernels + launch code,
// qp_attn_s6.hip: software-pipelined kernel).  Not part of the C ABI.
#pragma once
#include "qp_common.h"

namespace qpattn {

constexpr int kQB = 128;     // query rows per workgroup of the 4-wave kernels (s4, v1, s6<4>); s6<8> uses 256 (AttnParams::qb_rows)
constexpr int kKV = 64;      // keys per tile
constexpr int kD = 128;
constexpr int kPartialFloats = 4 * 64 * 64 + 4 * 2 * 64;   // per split of a 128-row item: O^T raw accumulators [wave][reg][lane] + (m,l) [wave][2][lane]
constexpr int partial_floats(int qb_rows) { return (qb_rows / 32) * (64 * 64 + 2 * 64); }   // ... of a qb_rows-row item

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

struct AttnParams {
  const uint4* q; uint2* out;
  const uint4* kp; const uint4* vp; int64_t pre_hs16; int64_t P;
  const uint4* kn; const uint4* vn; int64_t new_hs16; int64_t n;
  int hq; int group; float c;   // c = scale * log2(e)
  int nqb; int hkv;
  int items;                    // q-block x q-head-in-group items per kv head
  int n_whole;                  // first n_whole items of a kv head run unsplit; the rest are cut into `nsplit` kv ranges
  int nsplit;
  float* ws;                    // partial results of split items (kPartialFloats floats each)
  // batched non-causal mode (ViT tower: D = 80, one sequence per temporal patch, q/k/v interleaved in one qkv row):
  int heads_per_seq;            // "kv head" index = seq * heads_per_seq + head
  int64_t seq_stride16;         // uint4 between consecutive sequences (K/V and Q)
  int kv_row_bytes;             // byte stride between consecutive K/V (and Q) rows
  const int* cu_seqlens;        // ragged batch (Qwen2.5-VL window attention): sequence s = rows [cu[s], cu[s+1]); NULL = n_seq x S
  // query sub-range (group-token parallel ranks): q/out hold rows [q_row0, q_row0+nq) of the group's n new tokens
  int q_row0; int nq;
  int qb_rows;                  // query rows per workgroup / work item (128 or 256); partials hold partial_floats(qb_rows) floats
  // "flat" split (round 5; stream-K for attention): per kv head the items' key tiles are laid end to end, heaviest item first, and cut into
  // `flat_pieces` equal ranges — one workgroup per range, whatever item boundary falls inside it.  A workgroup walks its range segment by
  // segment (segment = the part of ONE item inside the range): a segment that covers its whole item writes the output, any other one
  // leaves a partial in slot (kv head, piece, first segment ? 0 : 1) for attn_combine_flat_kernel.  0 = the n_whole / nsplit scheme above.
  int flat_pieces;
  long long flat_total;         // key tiles of all items of one kv head (sum of attn_item_tiles)
  int variant;                  // host side only: the developer switch attn_variant this launch was planned under
  int prio_mode;                // s6, 8-wave form: 1 = s_setprio 1 for waves 4-7 (the younger half), 2 = for waves 0-3, 0 = none
};

// key tiles item `item` walks: the prefix tiles + the causal tiles up to the last row of its q block (the same count the kernels use)
__host__ __device__ inline int attn_qb_tiles(const AttnParams& p, int qb) {
  int blk_end = qb * p.qb_rows + p.qb_rows;
  if (blk_end > p.nq) blk_end = p.nq;
  blk_end += p.q_row0;
  return (int)((p.P + kKV - 1) / kKV) + (blk_end + kKV - 1) / kKV;
}
// flat position f (in tiles, items heaviest = latest q block first, `group` items per q block) -> item index, *off = tiles into that item
__host__ __device__ inline int attn_flat_locate(const AttnParams& p, long long f, int* off) {
  int item = 0;
  for (int qb = p.nqb - 1; qb >= 0; --qb) {
    const int nt = attn_qb_tiles(p, qb);
    const long long cls = (long long)nt * p.group;
    if (f < cls) { *off = (int)(f % nt); return item + (int)(f / nt); }
    f -= cls; item += p.group;
  }
  *off = 0;
  return item;                                            // == p.items: past the end
}
__host__ __device__ inline long long attn_flat_item_start(const AttnParams& p, int item) {
  long long f = 0;
  const int cls_of = item / p.group;
  for (int c = 0; c < cls_of; ++c) f += (long long)attn_qb_tiles(p, p.nqb - 1 - c) * p.group;
  return f + (long long)attn_qb_tiles(p, p.nqb - 1 - cls_of) * (item % p.group);
}
__host__ __device__ inline long long attn_flat_bound(const AttnParams& p, int j) { return (long long)j * p.flat_total / p.flat_pieces; }

__device__ __forceinline__ bf16x8_t lds_read_b128(const unsigned char* lds, int off) {
  return *reinterpret_cast<const bf16x8_t*>(lds + off);
}
__device__ __forceinline__ s16x4_t lds_read_tr16(const unsigned char* lds, int off) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + off));
}
__device__ __forceinline__ float xhalf_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace qpattn

// qp_attn_s6.hip
void qp_launch_attn_s6(const qpattn::AttnParams& p, bool xcd, unsigned per_kvh, hipStream_t s);   // p.qb_rows selects 4 or 8 waves
// qp_attn_s7.hip (256-row items, 4 waves x 64 rows, one wave per SIMD)
void qp_launch_attn_s7(const qpattn::AttnParams& p, bool xcd, unsigned per_kvh, hipStream_t s);
