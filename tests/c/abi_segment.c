/* A plain C host drives the WHOLE decoder-layer loop through include/quickprefill.h — no torch, no Python — twice:
 *   (a) ONE qp_prefill_segment call for 2 layers x (a 320-row group that prunes to 160 rows, then a 24-row tail that does not prune);
 *   (b) the same launches issued operator by operator (seams 1-4: qp_add_rmsnorm, qp_linear_act, qp_rope_append[_keys], qp_prefill_attn,
 *       qp_prune_keys, qp_swiglu, qp_add_inplace), the way a reference maintainer would bind them one seam at a time.
 * Both must leave bit-identical hidden rows, kept-index lists, cache lengths and cache rows.  Built and run by
 * tests/test_gpu_ops.py::test_c_abi_segment_from_a_plain_c_program. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "quickprefill.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_QP(x) do { int r_ = (x); if (r_ != QP_OK) { printf("qp error %d at line %d: %s\n", r_, __LINE__, qp_last_error()); return 3; } } while (0)

enum { L = 2, DM = 512, HQ = 4, HKV = 2, D = 128, I = 1024, CAP = 512, NG = 320, KEEP = 160, NT = 24 };

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static uint32_t lcg_state = 2024u;
static float rnd(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)((lcg_state >> 8) & 0xffff) / 65536.0f - 0.5f; }

static void* dev_bf16(size_t n, float scale, float offset) {
  uint16_t* h = malloc(n * 2);
  for (size_t i = 0; i < n; ++i) h[i] = f2bf(offset + scale * rnd());
  void* d = NULL;
  if (hipMalloc(&d, n * 2) != hipSuccess || hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice) != hipSuccess) { free(h); return NULL; }
  free(h);
  return d;
}
static void* dev_raw(size_t bytes) { void* d = NULL; if (hipMalloc(&d, bytes) != hipSuccess) return NULL; (void)hipMemset(d, 0, bytes); return d; }

typedef struct { void *h, *x, *qkv, *q, *att, *o, *gu, *act, *dn, *ks, *vs, *keys, *idx, *cos, *sin, *aws, *gws; size_t aws_b, gws_b; } bufs;

int main(void) {
  qp_ctx* ctx = NULL;
  CHECK_QP(qp_create(&ctx, 0));
  const int64_t QKV = (HQ + 2 * HKV) * D;
  qp_layer lay[2][L];                                   /* two copies of the model: one cache per run */
  for (int l = 0; l < L; ++l) {
    qp_layer w;
    w.ln1 = dev_bf16(DM, 0.2f, 1.0f); w.ln2 = dev_bf16(DM, 0.2f, 1.0f);
    w.w_qkv = dev_bf16((size_t)QKV * DM, 0.08f, 0.f); w.b_qkv = dev_bf16(QKV, 0.2f, 0.f);
    w.w_o = dev_bf16((size_t)DM * HQ * D, 0.08f, 0.f);
    w.w_gate_up = dev_bf16((size_t)2 * I * DM, 0.08f, 0.f); w.w_down = dev_bf16((size_t)DM * I, 0.06f, 0.f);
    for (int r = 0; r < 2; ++r) {
      lay[r][l] = w;
      lay[r][l].k_cache = dev_raw((size_t)HKV * CAP * D * 2); lay[r][l].v_cache = dev_raw((size_t)HKV * CAP * D * 2);
    }
  }
  void* emb_g = dev_bf16((size_t)NG * DM, 1.0f, 0.f);
  void* emb_t = dev_bf16((size_t)NT * DM, 1.0f, 0.f);
  const size_t nmax = NG;
  bufs b[2];
  for (int r = 0; r < 2; ++r) {
    b[r].h = dev_raw(nmax * DM * 2); b[r].x = dev_raw(nmax * DM * 2); b[r].qkv = dev_raw(nmax * QKV * 2); b[r].q = dev_raw(nmax * HQ * D * 2);
    b[r].att = dev_raw(nmax * HQ * D * 2); b[r].o = dev_raw(nmax * DM * 2); b[r].gu = dev_raw(nmax * 2 * I * 2); b[r].act = dev_raw(nmax * I * 2);
    b[r].dn = dev_raw(nmax * DM * 2); b[r].ks = dev_raw((size_t)HKV * nmax * D * 2); b[r].vs = dev_raw((size_t)HKV * nmax * D * 2);
    b[r].keys = dev_raw(nmax * 2); b[r].idx = dev_raw((size_t)L * nmax * 4); b[r].cos = dev_raw(nmax * (D / 2) * 2); b[r].sin = dev_raw(nmax * (D / 2) * 2);
    b[r].aws_b = 64u << 20; b[r].aws = dev_raw(b[r].aws_b); b[r].gws_b = 128u << 20; b[r].gws = dev_raw(b[r].gws_b);
  }
  int64_t len[2][L] = {{0, 0}, {0, 0}};
  const int32_t sections[3] = {16, 24, 24};
  const float eps = 1e-6f, scale = 0.08838834764831845f;
  int64_t* dpos = dev_raw(3 * nmax * 8);
  int64_t hpos[3 * NG];

  for (int segi = 0; segi < 2; ++segi) {                 /* segment 0: the pruning group; segment 1: the prompt tail */
    const int64_t n = segi == 0 ? NG : NT, kk = segi == 0 ? KEEP : -1, pos0 = segi == 0 ? 0 : NG;
    for (int st = 0; st < 3; ++st) for (int64_t t = 0; t < n; ++t) hpos[st * n + t] = pos0 + t;
    CHECK_HIP(hipMemcpy(dpos, hpos, 3 * n * 8, hipMemcpyHostToDevice));
    for (int r = 0; r < 2; ++r) {
      CHECK_QP(qp_mrope_table(ctx, dpos, n, sections, 1000000.0f, D, b[r].cos, b[r].sin, NULL));
      CHECK_HIP(hipMemcpy(b[r].h, segi == 0 ? emb_g : emb_t, (size_t)n * DM * 2, hipMemcpyDeviceToDevice));
    }
    /* (a) one call */
    {
      qp_segment g; memset(&g, 0, sizeof g);
      g.n_layers = L; g.hidden = DM; g.n_q_heads = HQ; g.n_kv_heads = HKV; g.head_dim = D; g.intermediate = I; g.rms_eps = eps; g.attn_scale = scale;
      g.n = n; g.cache_capacity = CAP; g.prune_mode = QP_PRUNE_KEY_NORMS_SMALL; g.attend_prefix = 1;
      g.h = b[0].h; g.x = b[0].x; g.qkv = b[0].qkv; g.q = b[0].q; g.att = b[0].att; g.o = b[0].o; g.gate_up = b[0].gu; g.act = b[0].act; g.down = b[0].dn;
      g.k_stage = b[0].ks; g.v_stage = b[0].vs; g.norm_keys = b[0].keys; g.kept_idx = b[0].idx; g.kept_idx_stride = nmax; g.cos = b[0].cos; g.sin = b[0].sin;
      g.attn_ws = b[0].aws; g.attn_ws_bytes = b[0].aws_b; g.gemm_ws = b[0].gws; g.gemm_ws_bytes = b[0].gws_b;
      int64_t keep[L] = {kk, kk};
      CHECK_QP(qp_prefill_segment(ctx, &g, lay[0], len[0], keep, NULL));
    }
    /* (b) operator by operator */
    {
      bufs* p = &b[1];
      const void* delta = NULL;
      for (int l = 0; l < L; ++l) {
        qp_layer* w = &lay[1][l];
        const int64_t past = len[1][l], hs = (int64_t)CAP * D;
        CHECK_QP(qp_add_rmsnorm(ctx, p->h, delta, w->ln1, p->x, n, DM, eps, NULL));
        CHECK_QP(qp_linear_act(ctx, p->x, w->w_qkv, w->b_qkv, 0, 1.0f, p->qkv, n, QKV, DM, 0, p->gws, p->gws_b, NULL));
        const void *kn, *vn; int64_t ns;
        if (kk >= 0) {
          CHECK_QP(qp_rope_append_keys(ctx, p->qkv, p->cos, p->sin, n, HQ, HKV, D, p->q, p->ks, p->vs, n * D, 0, NULL, p->keys, QP_PRUNE_KEY_NORMS_SMALL, NULL));
          kn = p->ks; vn = p->vs; ns = n * D;
        } else {
          CHECK_QP(qp_rope_append(ctx, p->qkv, p->cos, p->sin, n, HQ, HKV, D, p->q, w->k_cache, w->v_cache, hs, past, NULL, NULL));
          kn = (char*)w->k_cache + (size_t)past * D * 2; vn = (char*)w->v_cache + (size_t)past * D * 2; ns = hs;
        }
        CHECK_QP(qp_prefill_attn(ctx, p->q, w->k_cache, w->v_cache, hs, past, kn, vn, ns, n, HQ, HKV, D, scale, p->att, p->aws, p->aws_b, NULL));
        CHECK_QP(qp_linear_act(ctx, p->att, w->w_o, NULL, 0, 1.0f, p->o, n, DM, HQ * D, 0, p->gws, p->gws_b, NULL));
        if (kk >= 0) {
          CHECK_QP(qp_prune_keys(ctx, p->keys, n, kk, p->ks, p->vs, n * D, HKV, D, w->k_cache, w->v_cache, hs, past, (int32_t*)p->idx + (size_t)l * nmax, NULL));
          len[1][l] = past + kk;
        } else len[1][l] = past + n;
        CHECK_QP(qp_add_rmsnorm(ctx, p->h, p->o, w->ln2, p->x, n, DM, eps, NULL));
        CHECK_QP(qp_linear_act(ctx, p->x, w->w_gate_up, NULL, 0, 1.0f, p->gu, n, 2 * I, DM, 0, p->gws, p->gws_b, NULL));
        CHECK_QP(qp_swiglu(ctx, p->gu, n, I, p->act, NULL));
        CHECK_QP(qp_linear_act(ctx, p->act, w->w_down, NULL, 0, 1.0f, p->dn, n, DM, I, 0, p->gws, p->gws_b, NULL));
        delta = p->dn;
      }
      CHECK_QP(qp_add_inplace(ctx, p->h, delta, n * DM, NULL));
    }
    CHECK_HIP(hipDeviceSynchronize());
    /* compare */
    size_t hb = (size_t)n * DM * 2;
    uint16_t *h0 = malloc(hb), *h1 = malloc(hb);
    CHECK_HIP(hipMemcpy(h0, b[0].h, hb, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(h1, b[1].h, hb, hipMemcpyDeviceToHost));
    int finite = 1;
    for (size_t i = 0; i < (size_t)n * DM; ++i) if ((h0[i] & 0x7f80u) == 0x7f80u) finite = 0;
    if (memcmp(h0, h1, hb) != 0 || !finite) { printf("segment %d: hidden rows differ (or are not finite)\n", segi); return 1; }
    free(h0); free(h1);
    for (int l = 0; l < L; ++l) {
      if (len[0][l] != len[1][l]) { printf("segment %d layer %d: cache_len %lld vs %lld\n", segi, l, (long long)len[0][l], (long long)len[1][l]); return 1; }
      if (kk >= 0) {
        int32_t i0[KEEP], i1[KEEP];
        CHECK_HIP(hipMemcpy(i0, (int32_t*)b[0].idx + (size_t)l * nmax, KEEP * 4, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(i1, (int32_t*)b[1].idx + (size_t)l * nmax, KEEP * 4, hipMemcpyDeviceToHost));
        if (memcmp(i0, i1, sizeof i0) != 0) { printf("segment %d layer %d: kept lists differ\n", segi, l); return 1; }
        for (int j = 1; j < KEEP; ++j) if (i0[j] <= i0[j - 1] || i0[j] >= NG) { printf("kept list not ascending / out of range\n"); return 1; }
      }
      size_t cb = (size_t)HKV * CAP * D * 2;
      uint16_t *c0 = malloc(cb), *c1 = malloc(cb);
      CHECK_HIP(hipMemcpy(c0, lay[0][l].k_cache, cb, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(c1, lay[1][l].k_cache, cb, hipMemcpyDeviceToHost));
      for (int hh = 0; hh < HKV; ++hh)
        if (memcmp(c0 + (size_t)hh * CAP * D, c1 + (size_t)hh * CAP * D, (size_t)len[0][l] * D * 2) != 0) { printf("segment %d layer %d: cache rows differ\n", segi, l); return 1; }
      free(c0); free(c1);
    }
  }
  printf("cache_len %lld %lld\n", (long long)len[0][0], (long long)len[0][1]);
  if (len[0][0] != KEEP + NT || len[0][1] != KEEP + NT) { printf("unexpected cache lengths\n"); return 1; }
  printf("one-call segment == operator-by-operator: hidden rows, kept lists, cache rows bit-identical\n");
  qp_destroy(ctx);
  return 0;
}
