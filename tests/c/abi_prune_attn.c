/* A plain C caller of the drop-in boundary (include/quickprefill.h): no torch, no Python, only the HIP runtime for device memory.
 * Built and run by tests/test_gpu_ops.py::test_c_abi_from_a_plain_c_program (hipcc -x c, links libquickprefill.so).
 *
 *  seam 1  qp_prune_tail on a [4][512][128] bf16 K/V arena: 256 new rows behind 100 kept ones, keep the 128 smallest key norms.  Row t of
 *          the new segment is a fixed pattern scaled by 1 + 0.02*perm(t) (steps of 2 % >> the bf16 ulp of the norm), so the kept set is
 *          known without re-stating the kernel's summation order: the 128 rows with the smallest perm, in ascending position, and the
 *          compacted rows must be bit-for-bit copies.
 *  seam 3  qp_prefill_attn with ONE key visible to query 0 (n = 1, no prefix): softmax over one key is 1, the output row equals V.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "quickprefill.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_QP(x) do { int r_ = (x); if (r_ != QP_OK) { printf("qp error %d at line %d: %s\n", r_, __LINE__, qp_last_error()); return 3; } } while (0)

static uint16_t f2bf(float f) {           /* round to nearest even */
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

int main(void) {
  enum { HKV = 4, CAP = 512, D = 128, PAST = 100, N = 256, K = 128, HQ = 28 };
  const size_t elems = (size_t)HKV * CAP * D;
  uint16_t *hk = malloc(elems * 2), *hv = malloc(elems * 2), *ok = malloc(elems * 2), *ov = malloc(elems * 2);
  int perm[N];
  for (int t = 0; t < N; ++t) perm[t] = (t * 89 + 17) % N;          /* 89 is coprime with 256: a permutation */
  uint32_t lcg = 12345u;
  for (int h = 0; h < HKV; ++h)
    for (int r = 0; r < CAP; ++r)
      for (int d = 0; d < D; ++d) {
        lcg = lcg * 1664525u + 1013904223u;
        float base = ((float)((lcg >> 9) & 0xffff) / 65536.0f - 0.5f);
        float pat = 0.25f + 0.5f * (float)((d * 7 + h * 3) % 11) / 11.0f;      /* same pattern for every new row */
        float kvv = (r >= PAST && r < PAST + N) ? pat * (1.0f + 0.02f * (float)perm[r - PAST]) : base;
        hk[((size_t)h * CAP + r) * D + d] = f2bf(kvv);
        hv[((size_t)h * CAP + r) * D + d] = f2bf(base);
      }
  qp_ctx* ctx = NULL;
  CHECK_QP(qp_create(&ctx, 0));
  printf("library %s, %d CUs\n", qp_version(), qp_device_cus(ctx));
  void *dk, *dv, *dws; int32_t* didx;
  const size_t ws_bytes = qp_prune_workspace_bytes(N, K, HKV, D);
  CHECK_HIP(hipMalloc(&dk, elems * 2)); CHECK_HIP(hipMalloc(&dv, elems * 2));
  CHECK_HIP(hipMalloc(&dws, ws_bytes ? ws_bytes : 16)); CHECK_HIP(hipMalloc((void**)&didx, K * 4));
  CHECK_HIP(hipMemcpy(dk, hk, elems * 2, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dv, hv, elems * 2, hipMemcpyHostToDevice));
  CHECK_QP(qp_prune_tail(ctx, dk, dv, (int64_t)CAP * D, PAST, N, K, HKV, D, didx, QP_PRUNE_KEY_NORMS_SMALL, dws, ws_bytes, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  int32_t idx[K];
  CHECK_HIP(hipMemcpy(idx, didx, K * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(ok, dk, elems * 2, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(ov, dv, elems * 2, hipMemcpyDeviceToHost));
  int bad = 0, j = 0;
  for (int t = 0; t < N; ++t)
    if (perm[t] < K) { if (j >= K || idx[j] != t) ++bad; ++j; }                 /* kept = the K smallest scales, ascending position */
  if (j != K) ++bad;
  for (int h = 0; h < HKV && !bad; ++h) {
    if (memcmp(ok + (size_t)h * CAP * D, hk + (size_t)h * CAP * D, (size_t)PAST * D * 2)) ++bad;       /* the past is untouched */
    for (int i = 0; i < K; ++i) {
      const size_t dst = ((size_t)h * CAP + PAST + i) * D, src = ((size_t)h * CAP + PAST + idx[i]) * D;
      if (memcmp(ok + dst, hk + src, D * 2) || memcmp(ov + dst, hv + src, D * 2)) ++bad;
    }
  }
  printf("seam 1 (qp_prune_tail): %s\n", bad ? "MISMATCH" : "kept indices and compacted rows exact");

  /* error behaviour: batch-style misuse is rejected before any launch, with a message */
  int rc = qp_prune_tail(ctx, dk, dv, (int64_t)CAP * D, PAST, N, N + 1, HKV, D, didx, QP_PRUNE_KEY_NORMS_SMALL, dws, ws_bytes, NULL);
  if (rc == QP_OK || !qp_last_error()[0]) { printf("k > n was accepted\n"); ++bad; }

  /* seam 3: one query, one key */
  uint16_t hq[HQ * D], ho[HQ * D];
  for (int i = 0; i < HQ * D; ++i) hq[i] = f2bf((float)(i % 13) * 0.1f - 0.6f);
  void *dq, *dout, *daws;
  const size_t aws = qp_attn_workspace_bytes(ctx, 1, 0, HQ, HKV);
  CHECK_HIP(hipMalloc(&dq, sizeof hq)); CHECK_HIP(hipMalloc(&dout, sizeof ho)); CHECK_HIP(hipMalloc(&daws, aws ? aws : 16));
  CHECK_HIP(hipMemcpy(dq, hq, sizeof hq, hipMemcpyHostToDevice));
  CHECK_QP(qp_prefill_attn(ctx, dq, NULL, NULL, 0, 0, dk, dv, (int64_t)CAP * D, 1, HQ, HKV, D, 0.088388f, dout, daws, aws, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
  int bad3 = 0;
  for (int q = 0; q < HQ; ++q)
    if (memcmp(ho + q * D, ov + (size_t)(q / (HQ / HKV)) * CAP * D, D * 2)) ++bad3;            /* out[q] = V[kv head of q][row 0] */
  printf("seam 3 (qp_prefill_attn, one key): %s\n", bad3 ? "MISMATCH" : "output row equals the value row");
  qp_destroy(ctx);
  return (bad || bad3) ? 1 : 0;
}
