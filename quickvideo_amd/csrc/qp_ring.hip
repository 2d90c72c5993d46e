// Frame ring of the overlap producer (SURVEY §8 a11 / §8(b) "Threading"): a NATIVE producer thread fills pinned host slots,
// enqueues the H2D copies on a dedicated HIP stream and signals the consumer with events — no Python thread, no polling.
//
// Reference shape (lvu/models/qwen25_lvu_interleaved.py:237-342): a daemon Python thread calls next(vr), runs the HF processor under
// the GIL and put()s into a Queue(maxsize=3); the main thread polls it every 10 ms (:853-871).  Here the thread is a std::thread of
// this library; the frame source is a C callback that writes STRAIGHT into the pinned slot (a Python reader pays the GIL only
// inside its own next(); the built-in raw-file source of pre-decoded videos never touches the interpreter), and both directions of
// slot reuse are ordered by events:
//   host slot  s : may be refilled once the previous H2D copy out of it has finished      (producer thread waits on h2d_done[s])
//   device slot s: may be overwritten once the consumer's last GPU read of it has finished (copy stream waits on read_done[s])
// All buffers are the caller's (torch's pinned / device allocator); the ring owns its events and its thread.
// device < 0 (no qp_ctx): host-only ring — no HIP call is made and acquire() hands out the host slot; used by callers without a GPU
// (the CPU test double of the pipeline) and by the symbol / threading tests of the CPU suite.
#include "qp_common.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
using clk = std::chrono::steady_clock;
inline double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }
// create / destroy run on the CALLER's thread: they make the ring's device current for their HIP calls and put the caller's back
struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    else prev = -1;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

struct qp_frame_ring {
  int device = -1;                       // HIP ordinal, -1 = host-only
  int depth = 0;
  size_t slot_bytes = 0;
  std::vector<void*> host, dev;
  hipStream_t copy_stream = nullptr;
  std::vector<hipEvent_t> h2d_done, read_done;
  std::vector<char> h2d_valid, read_valid, slot_free;
  std::vector<int64_t> slot_group;       // group a slot currently holds (-1 none)
  std::vector<size_t> slot_fill;         // bytes of that group
  hipEvent_t origin = nullptr;           // caller's timing event (recorded before the first acquire): h2d_ms are relative to it

  std::mutex mu;
  std::mutex join_mu;                    // stop() / h2d_ms() / destroy() from different threads: one of them joins
  std::condition_variable cv;
  int64_t n_groups = 0, produced = 0;
  bool cancelled = false, started = false, finished = false;
  int error = 0;
  std::string error_msg;
  std::thread th;

  qp_frame_source_fn fn = nullptr;
  void* user = nullptr;
  // built-in source: frames of a raw uint8 file (a pre-decoded .npy video), picked by index
  int fd = -1;
  int64_t file_off = 0, frame_bytes = 0;
  int frames_per_group = 0, io_threads = 1;
  std::vector<int64_t> frame_idx;

  double t_busy = 0, t_wait_slot = 0, t_wait_h2d = 0, t_copy = 0;
  std::vector<float> h2d_ms;             // per group: H2D finished, ms after `origin` (NaN until known)
  // One timed event PER GROUP (recorded on the copy stream right behind the copy), kept until qp_frame_ring_h2d_ms resolves them all
  // against `origin` — a slot's own event is re-recorded `depth` groups later, so a stamp that could not be taken at that moment
  // (origin set late, or not complete yet) used to be lost for good and the group's frame wait was booked to the ViT.
  // Videos of more than kMaxStampEvents groups fall back to the per-slot stamps.
  static constexpr int64_t kMaxStampEvents = 8192;
  std::vector<hipEvent_t> stamp_ev;

  void fail(int status, const std::string& msg) {
    std::lock_guard<std::mutex> lk(mu);
    if (!error) { error = status; error_msg = msg; }
    finished = true;
    cv.notify_all();
  }
};

static bool ring_stamp(qp_frame_ring* r, int slot) {
  // timestamp of the H2D that last used `slot` (its event has completed or is being waited for by the caller)
  hipEvent_t origin;
  {
    std::lock_guard<std::mutex> lk(r->mu);
    origin = r->origin;
  }
  const int64_t g = r->slot_group[slot];
  if (g < 0 || !r->h2d_valid[slot] || !origin || g >= (int64_t)r->h2d_ms.size()) return true;
  float ms = NAN;
  if (hipEventElapsedTime(&ms, origin, r->h2d_done[slot]) == hipSuccess) r->h2d_ms[g] = ms;
  else (void)hipGetLastError();
  return true;
}

static int64_t file_source(void* user, int64_t g, void* dst, size_t capacity) {
  qp_frame_ring* r = (qp_frame_ring*)user;
  const int64_t f0 = g * r->frames_per_group;
  const int64_t f1 = std::min<int64_t>(f0 + r->frames_per_group, (int64_t)r->frame_idx.size());
  if (f0 >= f1) return 0;
  if ((size_t)((f1 - f0) * r->frame_bytes) > capacity) return -2;
  std::atomic<int> bad{0};
  auto read_range = [&](int64_t a, int64_t b) {
    for (int64_t f = a; f < b; ++f) {
      char* out = (char*)dst + (f - f0) * r->frame_bytes;
      int64_t off = r->file_off + r->frame_idx[f] * r->frame_bytes, left = r->frame_bytes;
      while (left > 0) {
        ssize_t got = pread(r->fd, out, (size_t)left, (off_t)off);
        if (got <= 0) { bad.store(1); return; }
        out += got; off += got; left -= got;
      }
    }
  };
  const int64_t nf = f1 - f0;
  int nt = (int)std::min<int64_t>(std::max(r->io_threads, 1), nf);
  if (nt <= 1) read_range(f0, f1);
  else {
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(read_range, f0 + nf * t / nt, f0 + nf * (t + 1) / nt);
    read_range(f0, f0 + nf / nt);
    for (auto& th : pool) th.join();
  }
  if (bad.load()) return -3;
  return nf * r->frame_bytes;
}

static void producer_main(qp_frame_ring* r) {
  if (r->device >= 0 && hipSetDevice(r->device) != hipSuccess) { r->fail(QP_ERR_HIP, "frame ring: hipSetDevice failed"); return; }
  double busy = 0, wait_slot = 0, wait_h2d = 0, copy = 0;     // this group's host seconds, added to the totals when it is published
  for (int64_t g = 0;; ++g) {
    {
      std::lock_guard<std::mutex> lk(r->mu);
      if (r->cancelled || g >= r->n_groups) break;
    }
    busy = wait_slot = wait_h2d = copy = 0;
    const int slot = (int)(g % r->depth);
    auto t0 = clk::now();
    auto wait_free = [&]() -> bool {                       // the consumer has released the group that held this slot
      std::unique_lock<std::mutex> lk(r->mu);
      r->cv.wait(lk, [&] { return r->slot_free[slot] || r->cancelled; });
      return !r->cancelled;
    };
    if (r->device < 0) {                                   // host-only: the consumer reads the host slot itself
      if (!wait_free()) break;
      wait_slot += secs(t0, clk::now());
    } else if (r->h2d_valid[slot]) {                       // previous copy out of the pinned slot must be done before it is overwritten
      hipError_t e = hipEventSynchronize(r->h2d_done[slot]);
      if (e != hipSuccess) { r->fail(QP_ERR_HIP, std::string("frame ring: hipEventSynchronize: ") + hipGetErrorString(e)); return; }
      ring_stamp(r, slot);
      wait_h2d += secs(t0, clk::now());
    }
    auto t1 = clk::now();
    const int64_t got = r->fn(r->user, g, r->host[slot], r->slot_bytes);
    auto t2 = clk::now();
    busy += secs(t1, t2);
    if (got == 0) {                                        // source ended early: groups >= g do not exist
      std::lock_guard<std::mutex> lk(r->mu);
      r->n_groups = g;
      break;
    }
    if (got < 0 || (size_t)got > r->slot_bytes) {
      r->fail(QP_ERR_INVALID, "frame ring: the frame source failed for group " + std::to_string(g) + " (returned " + std::to_string(got) + ")");
      return;
    }
    if (r->device >= 0) {
      if (!wait_free()) break;
      auto t3 = clk::now();
      wait_slot += secs(t2, t3);
      hipError_t e = hipSuccess;
      if (r->read_valid[slot]) e = hipStreamWaitEvent(r->copy_stream, r->read_done[slot], 0);
      if (e == hipSuccess) e = hipMemcpyAsync(r->dev[slot], r->host[slot], (size_t)got, hipMemcpyHostToDevice, r->copy_stream);
      if (e == hipSuccess) e = hipEventRecord(r->h2d_done[slot], r->copy_stream);
      if (e == hipSuccess && g < (int64_t)r->stamp_ev.size()) {      // the group's own timing event (best effort: a failure costs a stamp)
        if (hipEventCreate(&r->stamp_ev[g]) != hipSuccess || hipEventRecord(r->stamp_ev[g], r->copy_stream) != hipSuccess) {
          (void)hipGetLastError();
          if (r->stamp_ev[g]) { (void)hipEventDestroy(r->stamp_ev[g]); r->stamp_ev[g] = nullptr; }
        }
      }
      if (e != hipSuccess) { r->fail(QP_ERR_HIP, std::string("frame ring: H2D enqueue: ") + hipGetErrorString(e)); return; }
      r->h2d_valid[slot] = 1;
      copy += secs(t3, clk::now());
    }
    {
      std::lock_guard<std::mutex> lk(r->mu);
      r->slot_free[slot] = 0;
      r->slot_group[slot] = g;
      r->slot_fill[slot] = (size_t)got;
      r->read_valid[slot] = 0;
      r->produced = g + 1;
      r->t_busy += busy; r->t_wait_slot += wait_slot; r->t_wait_h2d += wait_h2d; r->t_copy += copy;
    }
    r->cv.notify_all();
  }
  {
    std::lock_guard<std::mutex> lk(r->mu);
    r->finished = true;
  }
  r->cv.notify_all();
}

extern "C" {

int qp_frame_ring_create(qp_ctx* ctx, int depth, size_t slot_bytes, void* const* host_slots, void* const* dev_slots, void* copy_stream,
                         qp_frame_ring** out) {
  QP_REQUIRE(out && host_slots, QP_ERR_INVALID, "qp_frame_ring_create: NULL argument");
  QP_REQUIRE(depth >= 1 && depth <= 64 && slot_bytes > 0, QP_ERR_INVALID, "qp_frame_ring_create: depth=%d slot_bytes=%zu", depth, slot_bytes);
  QP_REQUIRE((ctx != nullptr) == (dev_slots != nullptr), QP_ERR_INVALID,
             "qp_frame_ring_create: a device ring needs a context AND device slots; a host-only ring neither");
  for (int i = 0; i < depth; ++i)
    QP_REQUIRE(host_slots[i] && (!dev_slots || dev_slots[i]), QP_ERR_INVALID, "qp_frame_ring_create: slot %d is NULL", i);
  qp_frame_ring* r = nullptr;
  try {                                                   // allocation failures must not cross the C boundary (std::terminate)
    r = new qp_frame_ring();
    r->host.assign(host_slots, host_slots + depth);
    if (dev_slots) r->dev.assign(dev_slots, dev_slots + depth);
    r->h2d_valid.assign(depth, 0);
    r->read_valid.assign(depth, 0);
    r->slot_free.assign(depth, 1);
    r->slot_group.assign(depth, -1);
    r->slot_fill.assign(depth, 0);
    r->h2d_done.assign(depth, nullptr);
    r->read_done.assign(depth, nullptr);
  } catch (const std::exception& ex) {
    delete r;
    return qp_fail(QP_ERR_HIP, "qp_frame_ring_create: %s", ex.what());
  }
  r->device = ctx ? ctx->device : -1;
  r->depth = depth;
  r->slot_bytes = slot_bytes;
  r->copy_stream = (hipStream_t)copy_stream;
  if (r->device >= 0) {
    DeviceScope scope(r->device);
    hipError_t e = scope.ok ? hipSuccess : hipErrorInvalidDevice;
    for (int i = 0; i < depth && e == hipSuccess; ++i) {
      e = hipEventCreate(&r->h2d_done[i]);                                   // timed: h2d_ms
      if (e == hipSuccess) e = hipEventCreateWithFlags(&r->read_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
      for (auto ev : r->h2d_done) if (ev) (void)hipEventDestroy(ev);
      for (auto ev : r->read_done) if (ev) (void)hipEventDestroy(ev);
      delete r;
      return qp_fail(QP_ERR_HIP, "qp_frame_ring_create: %s", hipGetErrorString(e));
    }
  }
  *out = r;
  return QP_OK;
}

static int ring_start(qp_frame_ring* r, int64_t n_groups) {
  QP_REQUIRE(!r->started, QP_ERR_INVALID, "qp_frame_ring_start: the ring has been started already (one video per ring)");
  QP_REQUIRE(n_groups >= 0, QP_ERR_INVALID, "qp_frame_ring_start: n_groups=%lld", (long long)n_groups);
  try {
    r->n_groups = n_groups;
    r->h2d_ms.assign((size_t)n_groups, NAN);
    if (r->device >= 0 && n_groups <= qp_frame_ring::kMaxStampEvents) r->stamp_ev.assign((size_t)n_groups, nullptr);
    r->started = true;
    r->th = std::thread(producer_main, r);
  } catch (const std::exception& ex) {                    // bad_alloc / system_error must not cross the C boundary
    r->started = false;
    return qp_fail(QP_ERR_HIP, "qp_frame_ring_start: %s", ex.what());
  }
  return QP_OK;
}

int qp_frame_ring_start(qp_frame_ring* r, qp_frame_source_fn fn, void* user, int64_t n_groups) {
  QP_REQUIRE(r && fn, QP_ERR_INVALID, "qp_frame_ring_start: NULL argument");
  QP_REQUIRE(!r->started, QP_ERR_INVALID, "qp_frame_ring_start: the ring has been started already (one video per ring)");
  r->fn = fn;
  r->user = user;
  return ring_start(r, n_groups);
}

int qp_frame_ring_start_file(qp_frame_ring* r, const char* path, int64_t data_offset, int64_t frame_bytes, const int64_t* frame_idx,
                             int64_t n_frames, int frames_per_group, int io_threads) {
  QP_REQUIRE(r && path && (frame_idx || n_frames == 0), QP_ERR_INVALID, "qp_frame_ring_start_file: NULL argument");
  QP_REQUIRE(!r->started, QP_ERR_INVALID, "qp_frame_ring_start_file: the ring has been started already (one video per ring)");
  QP_REQUIRE(data_offset >= 0 && frame_bytes > 0 && n_frames >= 0 && frames_per_group > 0, QP_ERR_INVALID,
             "qp_frame_ring_start_file: offset=%lld frame_bytes=%lld n_frames=%lld frames_per_group=%d", (long long)data_offset,
             (long long)frame_bytes, (long long)n_frames, frames_per_group);
  QP_REQUIRE((size_t)frame_bytes * (size_t)frames_per_group <= r->slot_bytes, QP_ERR_INVALID,
             "qp_frame_ring_start_file: a group of %d frames x %lld bytes does not fit a %zu-byte slot", frames_per_group,
             (long long)frame_bytes, r->slot_bytes);
  for (int64_t i = 0; i < n_frames; ++i)
    QP_REQUIRE(frame_idx[i] >= 0, QP_ERR_INVALID, "qp_frame_ring_start_file: frame index %lld is negative", (long long)frame_idx[i]);
  int fd = open(path, O_RDONLY | O_CLOEXEC);
  QP_REQUIRE(fd >= 0, QP_ERR_INVALID, "qp_frame_ring_start_file: cannot open %s: %s", path, strerror(errno));
  r->fd = fd;
  r->file_off = data_offset;
  r->frame_bytes = frame_bytes;
  r->frames_per_group = frames_per_group;
  r->io_threads = io_threads < 1 ? 1 : (io_threads > 32 ? 32 : io_threads);
  try {
    r->frame_idx.assign(frame_idx, frame_idx + n_frames);
  } catch (const std::exception& ex) {
    close(fd);
    r->fd = -1;
    return qp_fail(QP_ERR_HIP, "qp_frame_ring_start_file: %s", ex.what());
  }
  r->fn = file_source;
  r->user = r;
  return ring_start(r, (n_frames + frames_per_group - 1) / frames_per_group);
}

int qp_frame_ring_set_origin(qp_frame_ring* r, void* origin_event) {
  QP_REQUIRE(r, QP_ERR_INVALID, "qp_frame_ring_set_origin: NULL ring");
  std::lock_guard<std::mutex> lk(r->mu);
  r->origin = (hipEvent_t)origin_event;
  return QP_OK;
}

int qp_frame_ring_acquire(qp_frame_ring* r, int64_t g, void* consumer_stream, void** ptr_out, size_t* bytes_out) {
  return qp_frame_ring_acquire_for(r, g, consumer_stream, -1, ptr_out, bytes_out);
}

int qp_frame_ring_acquire_for(qp_frame_ring* r, int64_t g, void* consumer_stream, int64_t timeout_ms, void** ptr_out, size_t* bytes_out) {
  QP_REQUIRE(r && ptr_out && bytes_out, QP_ERR_INVALID, "qp_frame_ring_acquire: NULL argument");
  QP_REQUIRE(r->started && g >= 0, QP_ERR_INVALID, "qp_frame_ring_acquire: group %lld of a ring that %s", (long long)g,
             r->started ? "was started" : "has not been started");
  int slot = (int)(g % r->depth);
  {
    std::unique_lock<std::mutex> lk(r->mu);
    auto ready = [&] { return r->produced > g || r->error || r->cancelled || r->finished; };
    if (timeout_ms < 0) r->cv.wait(lk, ready);
    else if (!r->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready))
      return QP_ERR_TIMEOUT;                               // not an error: nothing is written to qp_last_error, call again
    // a group that was published is delivered even if the source failed on a LATER one (the reference's queue hands out the items
    // in front of the exception, qwen25_lvu_interleaved.py:291-292)
    if (r->produced <= g && r->error) return qp_fail(r->error, "%s", r->error_msg.c_str());
    if (r->produced <= g)
      return qp_fail(QP_ERR_INVALID, r->cancelled ? "qp_frame_ring_acquire: the ring was stopped before group %lld"
                                                   : "qp_frame_ring_acquire: the source ended before group %lld", (long long)g);
    QP_REQUIRE(r->slot_group[slot] == g && !r->slot_free[slot], QP_ERR_INVALID,
               "qp_frame_ring_acquire: group %lld is no longer in the ring (its slot holds group %lld)", (long long)g, (long long)r->slot_group[slot]);
    *bytes_out = r->slot_fill[slot];
  }
  if (r->device >= 0) {
    hipError_t e = hipStreamWaitEvent((hipStream_t)consumer_stream, r->h2d_done[slot], 0);
    QP_REQUIRE(e == hipSuccess, QP_ERR_HIP, "qp_frame_ring_acquire: hipStreamWaitEvent: %s", hipGetErrorString(e));
    *ptr_out = r->dev[slot];
  } else {
    *ptr_out = r->host[slot];
  }
  return QP_OK;
}

int qp_frame_ring_mark_read(qp_frame_ring* r, int64_t g, void* consumer_stream) {
  QP_REQUIRE(r && g >= 0, QP_ERR_INVALID, "qp_frame_ring_mark_read: bad argument");
  const int slot = (int)(g % r->depth);
  {
    std::lock_guard<std::mutex> lk(r->mu);
    QP_REQUIRE(r->slot_group[slot] == g && !r->slot_free[slot], QP_ERR_INVALID, "qp_frame_ring_mark_read: group %lld is not held by the ring",
               (long long)g);
  }
  if (r->device < 0) return QP_OK;
  hipError_t e = hipEventRecord(r->read_done[slot], (hipStream_t)consumer_stream);
  QP_REQUIRE(e == hipSuccess, QP_ERR_HIP, "qp_frame_ring_mark_read: hipEventRecord: %s", hipGetErrorString(e));
  std::lock_guard<std::mutex> lk(r->mu);
  r->read_valid[slot] = 1;
  return QP_OK;
}

int qp_frame_ring_release(qp_frame_ring* r, int64_t g, void* consumer_stream) {
  QP_REQUIRE(r && g >= 0, QP_ERR_INVALID, "qp_frame_ring_release: bad argument");
  const int slot = (int)(g % r->depth);
  bool marked;
  {
    std::lock_guard<std::mutex> lk(r->mu);
    QP_REQUIRE(r->slot_group[slot] == g && !r->slot_free[slot], QP_ERR_INVALID, "qp_frame_ring_release: group %lld is not held by the ring",
               (long long)g);
    marked = r->read_valid[slot] != 0;
  }
  if (r->device >= 0 && !marked) {                          // no earlier mark_read: the last read is whatever the stream holds NOW
    int rc = qp_frame_ring_mark_read(r, g, consumer_stream);
    if (rc != QP_OK) return rc;
  }
  {
    std::lock_guard<std::mutex> lk(r->mu);
    r->slot_free[slot] = 1;
  }
  r->cv.notify_all();
  return QP_OK;
}

int qp_frame_ring_stop(qp_frame_ring* r) {
  QP_REQUIRE(r, QP_ERR_INVALID, "qp_frame_ring_stop: NULL ring");
  {
    std::lock_guard<std::mutex> lk(r->mu);
    r->cancelled = true;
  }
  r->cv.notify_all();
  std::lock_guard<std::mutex> jl(r->join_mu);
  if (r->th.joinable()) r->th.join();
  return QP_OK;
}

int qp_frame_ring_stats(qp_frame_ring* r, double* out, int n_out) {
  QP_REQUIRE(r && out && n_out >= 0, QP_ERR_INVALID, "qp_frame_ring_stats: bad argument");
  std::lock_guard<std::mutex> lk(r->mu);
  const double v[6] = {r->t_busy, r->t_wait_slot, r->t_wait_h2d, r->t_copy, (double)r->produced, (double)r->n_groups};
  for (int i = 0; i < n_out && i < 6; ++i) out[i] = v[i];
  return QP_OK;
}

int qp_frame_ring_h2d_ms(qp_frame_ring* r, float* out, int64_t n_out) {
  QP_REQUIRE(r && (out || n_out == 0) && n_out >= 0, QP_ERR_INVALID, "qp_frame_ring_h2d_ms: bad argument");
  {
    std::lock_guard<std::mutex> lk(r->mu);
    QP_REQUIRE(!r->started || r->finished || r->cancelled, QP_ERR_INVALID,
               "qp_frame_ring_h2d_ms: the producer is still running (read the timestamps after the last group was acquired)");
  }
  {
    std::lock_guard<std::mutex> jl(r->join_mu);
    if (r->th.joinable()) r->th.join();
  }
  if (r->device >= 0 && r->origin) {
    DeviceScope scope(r->device);
    for (int s = 0; s < r->depth; ++s)
      if (r->h2d_valid[s] && hipEventSynchronize(r->h2d_done[s]) == hipSuccess) ring_stamp(r, s);
    // the per-group events: every copy has finished by now (the slots' events above are the last ones on the copy stream); origin is
    // waited for explicitly so that "not ready" cannot cost a stamp
    if (!r->stamp_ev.empty() && hipEventSynchronize(r->origin) == hipSuccess)
      for (size_t g = 0; g < r->stamp_ev.size() && g < r->h2d_ms.size(); ++g) {
        float ms = NAN;
        if (r->stamp_ev[g] && hipEventElapsedTime(&ms, r->origin, r->stamp_ev[g]) == hipSuccess) r->h2d_ms[g] = ms;
        else (void)hipGetLastError();
      }
  }
  for (int64_t g = 0; g < n_out; ++g) out[g] = g < (int64_t)r->h2d_ms.size() ? r->h2d_ms[g] : NAN;
  return QP_OK;
}

void qp_frame_ring_destroy(qp_frame_ring* r) {
  if (!r) return;
  (void)qp_frame_ring_stop(r);
  if (r->device >= 0) {
    DeviceScope scope(r->device);
    for (int s = 0; s < r->depth; ++s) {
      if (r->h2d_valid[s]) (void)hipEventSynchronize(r->h2d_done[s]);      // the caller's buffers outlive the copies that read them
      if (r->h2d_done[s]) (void)hipEventDestroy(r->h2d_done[s]);
      if (r->read_done[s]) (void)hipEventDestroy(r->read_done[s]);
    }
    for (auto ev : r->stamp_ev) if (ev) (void)hipEventDestroy(ev);
  }
  if (r->fd >= 0) close(r->fd);
  delete r;
}

}  // extern "C"
