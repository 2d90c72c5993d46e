/* Plain-C99 caller of the frame ring (include/quickprefill.h: qp_frame_ring_*): what a cgo / JNI host of the overlap producer binds.
 * `abi_frame_ring host`   : host-only ring (no GPU): a C source callback, 3 slots, 11 groups with a short last one, every byte checked,
 *                           the source never runs further ahead than the ring allows, early end, error from the source, stop() while full.
 * `abi_frame_ring device` : the same stream of groups through pinned slots -> H2D on a copy stream -> device slots, read back on a
 *                           consumer stream that is slowed down by host sleeps; copy timestamps against an origin event. */
#define _DEFAULT_SOURCE            /* usleep under -std=c99 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "quickprefill.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, qp_last_error()); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define FRAME 4096
#define FPG 4
#define DEPTH 3

typedef struct { int64_t groups, last_frames, fail_at, end_at, max_ahead; volatile int64_t released; } source_t;

static unsigned char pixel(int64_t g, size_t i) { return (unsigned char)((g * 131 + (int64_t)i * 7 + (int64_t)(i >> 9)) & 255); }

static int64_t source(void* user, int64_t g, void* dst, size_t capacity) {
  source_t* s = (source_t*)user;
  if (g == s->fail_at) return -7;
  if (g == s->end_at) return 0;
  int64_t frames = g == s->groups - 1 ? s->last_frames : FPG;
  size_t bytes = (size_t)frames * FRAME, i;
  if (bytes > capacity) return -1;
  if (g - s->released > s->max_ahead) s->max_ahead = g - s->released;
  for (i = 0; i < bytes; ++i) ((unsigned char*)dst)[i] = pixel(g, i);
  return (int64_t)bytes;
}

static int check_bytes(const unsigned char* p, int64_t g, size_t bytes) {
  size_t i;
  for (i = 0; i < bytes; ++i) if (p[i] != pixel(g, i)) { fprintf(stderr, "group %lld byte %zu: %d != %d\n", (long long)g, i, p[i], pixel(g, i)); return 1; }
  return 0;
}

static int run_host(void) {
  void* host[DEPTH];
  int i; int64_t g;
  for (i = 0; i < DEPTH; ++i) host[i] = malloc(FPG * FRAME);
  {
    source_t s = {11, 2, -1, -1, 0, 0};
    qp_frame_ring* r = NULL;
    double st[6];
    CHECK(qp_frame_ring_create(NULL, DEPTH, FPG * FRAME, host, NULL, NULL, &r));
    CHECK(qp_frame_ring_start(r, source, &s, s.groups));
    if (qp_frame_ring_start(r, source, &s, s.groups) == 0) { fprintf(stderr, "second start accepted\n"); return 1; }
    for (g = 0; g < s.groups; ++g) {
      void* p; size_t bytes;
      if (g & 1) { CHECK(qp_frame_ring_acquire(r, g, NULL, &p, &bytes)); }
      else {                                               /* the bounded form, as an interruptible host would call it */
        int rc;
        while ((rc = qp_frame_ring_acquire_for(r, g, NULL, 5, &p, &bytes)) == QP_ERR_TIMEOUT) {}
        CHECK(rc);
      }
      if (p != host[g % DEPTH] || bytes != (size_t)(g == 10 ? 2 : FPG) * FRAME || check_bytes((unsigned char*)p, g, bytes)) return 1;
      if (g == 4) usleep(50000);                           /* a slow consumer: the source must stop at the ring's depth */
      s.released = g + 1;
      CHECK(qp_frame_ring_release(r, g, NULL));
      if (qp_frame_ring_release(r, g, NULL) == 0) { fprintf(stderr, "double release accepted\n"); return 1; }
    }
    CHECK(qp_frame_ring_stats(r, st, 6));
    if (st[4] != 11 || st[5] != 11 || s.max_ahead > DEPTH) { fprintf(stderr, "produced %g of %g, ran %lld ahead\n", st[4], st[5], (long long)s.max_ahead); return 1; }
    qp_frame_ring_destroy(r);
  }
  {                                                         /* early end at group 2, failure at group 1 */
    source_t s = {5, FPG, -1, 2, 0, 0};
    qp_frame_ring* r = NULL; void* p; size_t bytes;
    CHECK(qp_frame_ring_create(NULL, DEPTH, FPG * FRAME, host, NULL, NULL, &r));
    CHECK(qp_frame_ring_start(r, source, &s, s.groups));
    CHECK(qp_frame_ring_acquire(r, 1, NULL, &p, &bytes));
    if (qp_frame_ring_acquire(r, 2, NULL, &p, &bytes) == 0 || !strstr(qp_last_error(), "ended before group 2")) { fprintf(stderr, "early end: %s\n", qp_last_error()); return 1; }
    qp_frame_ring_destroy(r);
    s.end_at = -1; s.fail_at = 1;
    CHECK(qp_frame_ring_create(NULL, DEPTH, FPG * FRAME, host, NULL, NULL, &r));
    CHECK(qp_frame_ring_start(r, source, &s, s.groups));
    CHECK(qp_frame_ring_acquire(r, 0, NULL, &p, &bytes));
    if (qp_frame_ring_acquire(r, 1, NULL, &p, &bytes) == 0 || !strstr(qp_last_error(), "returned -7")) { fprintf(stderr, "source failure: %s\n", qp_last_error()); return 1; }
    qp_frame_ring_destroy(r);
  }
  {                                                         /* stop() while the producer waits on a full ring */
    source_t s = {50, FPG, -1, -1, 0, 0};
    qp_frame_ring* r = NULL; void* p; size_t bytes;
    CHECK(qp_frame_ring_create(NULL, DEPTH, FPG * FRAME, host, NULL, NULL, &r));
    CHECK(qp_frame_ring_start(r, source, &s, s.groups));
    CHECK(qp_frame_ring_acquire(r, 2, NULL, &p, &bytes));
    usleep(20000);
    CHECK(qp_frame_ring_stop(r));
    CHECK(qp_frame_ring_stop(r));
    if (qp_frame_ring_acquire(r, 3, NULL, &p, &bytes) == 0) { fprintf(stderr, "acquire after stop succeeded\n"); return 1; }
    qp_frame_ring_destroy(r);
  }
  for (i = 0; i < DEPTH; ++i) free(host[i]);
  printf("abi_frame_ring host: ok\n");
  return 0;
}

static int run_device(void) {
  qp_ctx* ctx = NULL; qp_frame_ring* r = NULL;
  void *host[DEPTH], *dev[DEPTH];
  unsigned char* back;
  hipStream_t copy, consumer; hipEvent_t origin;
  source_t s = {14, 3, -1, -1, 0, 0};
  float ms[14]; double st[6];
  int i; int64_t g;
  CHECK(qp_create(&ctx, 0));
  HIPCHECK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
  HIPCHECK(hipStreamCreateWithFlags(&consumer, hipStreamNonBlocking));
  HIPCHECK(hipEventCreate(&origin));
  for (i = 0; i < DEPTH; ++i) { HIPCHECK(hipHostMalloc(&host[i], FPG * FRAME, 0)); HIPCHECK(hipMalloc(&dev[i], FPG * FRAME)); }
  HIPCHECK(hipHostMalloc((void**)&back, 14 * FPG * FRAME, 0));
  CHECK(qp_frame_ring_create(ctx, DEPTH, FPG * FRAME, host, dev, copy, &r));
  HIPCHECK(hipEventRecord(origin, consumer));
  CHECK(qp_frame_ring_set_origin(r, origin));
  CHECK(qp_frame_ring_start(r, source, &s, s.groups));
  for (g = 0; g < s.groups; ++g) {
    void* p; size_t bytes;
    CHECK(qp_frame_ring_acquire(r, g, consumer, &p, &bytes));
    if (p != dev[g % DEPTH]) return 1;
    /* the consumer's "GPU read" of the slot: a device-to-host copy on its own stream, never synchronised inside the loop */
    HIPCHECK(hipMemcpyAsync(back + g * FPG * FRAME, p, bytes, hipMemcpyDeviceToHost, consumer));
    CHECK(qp_frame_ring_mark_read(r, g, consumer));
    s.released = g + 1;
    CHECK(qp_frame_ring_release(r, g, consumer));
  }
  HIPCHECK(hipStreamSynchronize(consumer));
  for (g = 0; g < s.groups; ++g) if (check_bytes(back + g * FPG * FRAME, g, (size_t)(g == 13 ? 3 : FPG) * FRAME)) return 1;
  CHECK(qp_frame_ring_stop(r));
  CHECK(qp_frame_ring_h2d_ms(r, ms, 14));
  for (g = 0; g < 14; ++g) if (!(ms[g] >= 0.0f) || (g && ms[g] < ms[g - 1])) { fprintf(stderr, "h2d_ms[%lld] = %f\n", (long long)g, ms[g]); return 1; }
  CHECK(qp_frame_ring_stats(r, st, 6));
  qp_frame_ring_destroy(r);
  qp_destroy(ctx);
  printf("abi_frame_ring device: ok (14 groups, last copy %.3f ms after the origin, source busy %.4f s)\n", ms[13], st[0]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "device") == 0) return run_device();
  return run_host();
}
