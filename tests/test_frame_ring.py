"""The native frame ring (include/quickprefill.h qp_frame_ring_*, csrc/qp_ring.hip, quickvideo_amd/ring.py): SURVEY §8 a11 /
§8(b) "Threading" — the reference's daemon thread + Queue(maxsize=3) + 10 ms polling (qwen25_lvu_interleaved.py:237-342, 853-871)
as a library thread with pinned slots, a copy stream and events.

CPU part (host-only ring: no HIP call): ordering, byte equality, back-pressure, early end, errors raised by a Python source arriving
in the consumer as the SAME exception, stop() on a full ring, the built-in file source, the plain-C caller.
GPU part: the same through device slots with a deliberately slow consumer stream, and the plugin's generate() giving the same tokens
whichever producer feeds it (native + reader callback, native + file source, the Python thread)."""
import os
import subprocess
import threading
import time

import numpy as np
import pytest
import torch

from quickvideo_amd.frames import open_video
from quickvideo_amd.native import QuickPrefillError
from quickvideo_amd.ring import FrameRing

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 28, 56


def _reader(F=22, gs=4, seed=3):
    r = open_video(f"synthetic://v?frames={F}&h={H}&w={W}&seed={seed}")
    r.height, r.width, r.frame_iter = H, W, gs
    r.process(range(F))
    return r


def _slots(n=3, gs=4):
    return [torch.zeros(gs, 3, H, W, dtype=torch.uint8) for _ in range(n)]


def test_reader_source_delivers_every_group_in_order_with_a_short_last_one():
    want = [f.clone() for f in _reader()]
    assert [w.shape[0] for w in want] == [4, 4, 4, 4, 4, 2]
    ring = FrameRing(_slots())
    ring.start_reader(_reader(), len(want))
    for g, w in enumerate(want):
        got = ring.acquire(g)
        assert got.shape == w.shape and torch.equal(got, w), g
        ring.release(g)
    st = ring.stats()
    assert st["produced"] == st["groups"] == 6 and st["busy"] > 0
    ring.close()


def test_next_into_writes_the_slot_itself():
    """Readers of the package decode INTO the pinned slot: the array handed to next_into is the slot's memory."""
    seen = []
    r = _reader()
    orig = r.next_into
    r.next_into = lambda dst: (seen.append(dst.ctypes.data), orig(dst))[1]
    slots = _slots()
    ring = FrameRing(slots)
    ring.start_reader(r, 6)
    for g in range(6):
        ring.acquire(g)
        ring.release(g)
    ring.close()
    assert seen == [slots[g % 3].data_ptr() for g in range(6)]


def test_foreign_reader_with_only_next_is_copied_in():
    want = [f.clone() for f in _reader()]

    class Foreign:                                   # the bare InterleavedVideoReader contract: __next__ -> uint8 [g,3,H,W]
        def __init__(self):
            self.it = iter(want)

        def __next__(self):
            return next(self.it).numpy()             # a numpy array, not a tensor: also accepted

    ring = FrameRing(_slots())
    ring.start_reader(Foreign(), len(want))
    for g, w in enumerate(want):
        assert torch.equal(ring.acquire(g), w)
        ring.release(g)
    ring.close()


def test_source_is_bounded_by_the_ring_depth():
    """Queue(maxsize=3) semantics: with nothing released the source is asked for exactly `depth` groups, then waits."""
    asked = []

    class Counting:
        def __next__(self):
            asked.append(len(asked))
            return torch.full((4, 3, H, W), len(asked) - 1, dtype=torch.uint8)

    ring = FrameRing(_slots(3))
    ring.start_reader(Counting(), 10)
    ring.acquire(2)
    time.sleep(0.15)
    assert asked == [0, 1, 2]
    ring.release(0)
    ring.acquire(3)
    time.sleep(0.1)
    assert asked == [0, 1, 2, 3] and int(ring.acquire(3)[0, 0, 0, 0]) == 3
    ring.close()                                     # stop() while the producer waits for a slot
    assert not [t for t in threading.enumerate() if "qp_frame" in t.name]


def test_python_exception_in_the_source_reaches_the_consumer():
    class Bad:
        def __init__(self):
            self.i = 0

        def __next__(self):
            self.i += 1
            if self.i == 2:
                raise KeyError("decoder exploded")
            return torch.zeros(4, 3, H, W, dtype=torch.uint8)

    ring = FrameRing(_slots())
    ring.start_reader(Bad(), 5)
    ring.acquire(0)
    ring.release(0)
    with pytest.raises(KeyError, match="decoder exploded"):
        ring.acquire(1)
    ring.close()


def test_wrong_shape_from_a_reader_is_an_error_not_a_corrupt_slot():
    class Wrong:
        def __next__(self):
            return torch.zeros(4, 3, H, W + 2, dtype=torch.uint8)

    ring = FrameRing(_slots())
    ring.start_reader(Wrong(), 2)
    with pytest.raises(ValueError, match="reader returned"):
        ring.acquire(0)
    ring.close()


def test_early_end_and_misuse_are_named_errors():
    ring = FrameRing(_slots())
    with pytest.raises(QuickPrefillError, match="has not been started"):
        ring.acquire(0)
    ring.start_reader(_reader(F=8), 5)               # the reader has 2 groups, the plan says 5
    ring.acquire(0), ring.release(0)
    ring.acquire(1), ring.release(1)
    with pytest.raises(QuickPrefillError, match="ended before group 2"):
        ring.acquire(2)
    with pytest.raises(QuickPrefillError, match="not held by the ring"):
        ring.release(1)
    with pytest.raises(QuickPrefillError, match="started already"):
        ring.start_reader(_reader(), 1)
    ring.close()
    ring.close()                                     # idempotent


def test_file_source_reads_a_npy_video_without_the_interpreter(tmp_path):
    arr = np.random.RandomState(0).randint(0, 256, (30, 3, H, W), dtype=np.uint8)
    np.save(tmp_path / "v.npy", arr)
    r = open_video(str(tmp_path / "v.npy"))
    path, off, frame_bytes = r.raw_layout()
    assert frame_bytes == 3 * H * W and open(path, "rb").read()[off:off + 16] == arr.tobytes()[:16]
    idx = [29, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26]           # any order, 15 frames: 4 groups, the last holds 3
    for threads in (1, 3):
        ring = FrameRing(_slots())
        ring.start_file(path, off, idx, 4, io_threads=threads)
        for g in range(4):
            assert np.array_equal(ring.acquire(g).numpy(), arr[idx[g * 4:(g + 1) * 4]]), (threads, g)
            ring.release(g)
        with pytest.raises(QuickPrefillError, match="ended before group 4"):
            ring.acquire(4)
        ring.close()
    ring = FrameRing(_slots())
    with pytest.raises(QuickPrefillError, match="cannot open"):
        ring.start_file(str(tmp_path / "missing.npy"), 0, [0], 4)
    ring.close()
    ring = FrameRing(_slots())
    ring.start_file(path, off, [0, 1, 2, 3, 4000], 4)                       # an index past the end of the file: a read error, reported
    ring.acquire(0)
    ring.release(0)
    with pytest.raises(QuickPrefillError, match="frame source failed for group 1"):
        ring.acquire(1)
    ring.close()


def test_bounded_acquire_times_out_and_leaves_the_ring_intact():
    """qp_frame_ring_acquire_for: a hung source costs the consumer one bounded wait at a time (the reference polls its queue every
    10 ms so Ctrl-C works, qwen25_lvu_interleaved.py:853-871) — QP_ERR_TIMEOUT is not an error: no message, the group arrives later."""
    import ctypes
    from quickvideo_amd.native import QP_ERR_TIMEOUT, QP_OK
    gate = threading.Event()

    class Slow:
        height, width, frame_iter = H, W, 4

        def __init__(self):
            self.inner = _reader()

        def __next__(self):
            gate.wait(30)
            return next(self.inner)

    ring = FrameRing(_slots())
    ring.start_reader(Slow(), 6)
    ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
    before = ring.lib.qp_last_error()
    t0 = time.perf_counter()
    assert ring.lib.qp_frame_ring_acquire_for(ring.h, 0, None, 60, ctypes.byref(ptr), ctypes.byref(nbytes)) == QP_ERR_TIMEOUT
    assert 0.05 <= time.perf_counter() - t0 < 5.0 and ring.lib.qp_last_error() == before
    ring.POLL_MS = 20
    threading.Timer(0.2, gate.set).start()
    got = ring.acquire(0)                                   # the Python face loops over bounded waits
    assert torch.equal(got, next(_reader()))
    assert ring.lib.qp_frame_ring_acquire_for(ring.h, 0, None, -1, ctypes.byref(ptr), ctypes.byref(nbytes)) == QP_OK   # < 0: wait for good
    ring.release(0)
    ring.close()


def test_native_file_source_is_taken_only_for_the_file_as_it_lies(tmp_path):
    """the library pread()s ring.frame_bytes per frame, so the built-in file source may only run when that IS the stored
    frame; a plan whose frame size differs goes through the reader (whose own check raises), a partly consumed selection hands over
    only what is left, and the reader's cursor is advanced."""
    from quickvideo_amd import pipeline
    arr = np.random.RandomState(0).randint(0, 256, (12, 3, H, W), dtype=np.uint8)
    np.save(tmp_path / "v.npy", arr)

    class FakeRing:
        def __init__(self, shape):
            self.frame_shape, self.frame_bytes, self.calls = shape, int(np.prod(shape)), []

        def start_file(self, path, off, idx, fpg, io_threads=8):
            self.calls.append(("file", [int(i) for i in idx]))

        def start_reader(self, reader, n_groups):
            self.calls.append(("reader", n_groups))

    def producer(reader, shape):
        p = object.__new__(pipeline._NativeProducer)
        p.reader, p.n_groups, p.fpg, p.native_file, p.ring = reader, 3, 4, False, FakeRing(shape)
        p.start()
        return p

    r = open_video(str(tmp_path / "v.npy"))
    r.frame_iter = 4
    r.process(range(12))
    next(r)                                                                   # one group already consumed by the caller
    p = producer(r, (3, H, W))
    assert p.native_file and p.ring.calls == [("file", list(range(4, 12)))] and len(r.pending_indices()) == 0
    r = open_video(str(tmp_path / "v.npy"))
    r.process(range(12))
    p = producer(r, (3, H, W // 2))                                           # the plan's frame is not the stored one
    assert not p.native_file and p.ring.calls == [("reader", 3)]
    r = open_video(str(tmp_path / "v.npy"))
    r.height, r.width = H, W // 2                                             # preset size the reader itself refuses
    r.process(range(12))
    assert r.raw_layout() is None
    with pytest.raises(ValueError, match="no resizer"):
        next(r)

    class Sub(type(r)):
        def _frames(self, idx, out=None):
            return super()._frames(idx, out)

    assert Sub(str(tmp_path / "v.npy")).raw_layout() is None                   # a subclass that makes its frames its own way


def test_stall_attribution_keeps_unknown_upload_times_apart():
    """pipeline._device_breakdown: the idle time in front of a group is split at the moment its frames finished uploading — frame wait
    before, ViT wait after.  A group whose upload time the library could not stamp (NaN) is booked to NEITHER (it used to read
    as 0.0 = "uploaded long ago", i.e. the whole stall went to the tower)."""
    from quickvideo_amd.pipeline import PrefillPipeline, Timings
    tm = Timings()
    #        h2d_done  vit0   vit1   prefill0 prefill1   (ms after origin; floats = the native ring's stamps)
    trace = [(5.0,     6.0,   9.0,   10.0,    20.0),      # group 0: 10 ms in front of it, frames there at 5 -> 5 frames + 5 ViT
             (31.0,    32.0,  33.0,  34.0,    40.0),      # group 1: prefill(0) ended at 20, frames at 31 -> 11 frames + 3 ViT
             (float("nan"), 41.0, 47.0, 48.0, 50.0)]      # group 2: 8 ms idle, upload time unknown -> unattributed
    PrefillPipeline._device_breakdown(tm, None, trace)
    assert tm.group_gaps == pytest.approx([0.010, 0.014, 0.008])
    assert tm.gpu_stall_frames == pytest.approx(0.016) and tm.gpu_stall_vit == pytest.approx(0.008)
    assert tm.gpu_stall_unknown == pytest.approx(0.008)
    assert tm.gpu_prefill_busy == pytest.approx(0.018) and tm.vit_span == pytest.approx(0.010)


def test_pt_video_has_no_raw_layout(tmp_path):
    torch.save(torch.zeros(4, 3, H, W, dtype=torch.uint8), tmp_path / "v.pt")
    assert open_video(str(tmp_path / "v.pt")).raw_layout() is None


def _c_caller(tmp_path):
    from tests.test_abi import build_c_caller
    exe = str(tmp_path / "abi_frame_ring")
    build_c_caller(exe, os.path.join(ROOT, "tests", "c", "abi_frame_ring.c"))
    return exe


def test_plain_c_caller_host_ring(tmp_path):
    p = subprocess.run([_c_caller(tmp_path), "host"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "abi_frame_ring host: ok" in p.stdout, p.stdout + p.stderr


# ------------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_plain_c_caller_device_ring(tmp_path):
    p = subprocess.run([_c_caller(tmp_path), "device"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "abi_frame_ring device: ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_device_ring_under_a_slow_consumer_stream():
    """12 groups through 3 device slots while the consumer stream is held back by a spin kernel in front of every read: the producer
    runs as far ahead as the ring lets it, so a slot overwritten before its read (missing read_done wait) or a pinned slot refilled
    before its copy (missing h2d wait) would show up as wrong bytes.  Copy timestamps: finite, ascending, after the origin."""
    from quickvideo_amd.native import QuickPrefillOps
    dev = torch.device("cuda:0")
    ops = QuickPrefillOps(dev)
    gs, F = 4, 46
    want = [f.clone() for f in _reader(F=F, gs=gs, seed=11)]
    host = [torch.empty(gs, 3, H, W, dtype=torch.uint8).pin_memory() for _ in range(3)]
    dslots = [torch.empty(gs, 3, H, W, dtype=torch.uint8, device=dev) for _ in range(3)]
    copy, cons = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ring = FrameRing(host, dslots, ctx=ops.ctx, copy_stream=copy)
    origin = torch.cuda.Event(enable_timing=True)
    origin.record(cons)
    ring.set_origin(origin)
    ring.start_reader(_reader(F=F, gs=gs, seed=11), len(want))
    outs = []
    for g in range(len(want)):
        frames = ring.acquire(g, cons)
        assert frames.is_cuda and frames.data_ptr() == dslots[g % 3].data_ptr()
        with torch.cuda.stream(cons):
            torch.cuda._sleep(10_000_000)                       # ~5 ms of spin in front of the read
            outs.append(frames.clone())
        ring.mark_read(g, cons)
        ring.release(g, cons)
    cons.synchronize()
    for g, w in enumerate(want):
        assert torch.equal(outs[g].cpu(), w), g
    ring.stop()
    ms = ring.h2d_ms(len(want))
    assert all(m == m and m >= 0 for m in ms) and ms == sorted(ms), ms
    st = ring.stats()
    assert st["produced"] == len(want) == st["groups"] and st["busy"] > 0 and st["copy"] > 0, st
    ring.close()


@pytest.mark.gpu
def test_generate_gives_the_same_tokens_whichever_producer_feeds_it(tmp_path, monkeypatch):
    """The plugin on a pre-decoded .npy video: native ring + built-in file source (default), native ring + reader callback, and the
    Python thread (QP_NATIVE_PRODUCER=0) — same tokens, and the timings say which producer ran."""
    import lvu
    from quickvideo_amd.lvu import load_native_model
    from quickvideo_amd import pipeline
    m = load_native_model("synthetic:tiny", device="cuda:0", seed=3)
    frames = np.random.RandomState(4).randint(0, 256, (40, 3, 112, 168), dtype=np.uint8)
    video = str(tmp_path / "v.npy")
    np.save(video, frames)
    made = []
    for cls in ("_NativeProducer", "_Producer"):
        orig = getattr(pipeline, cls).start

        def start(self, _orig=orig):
            made.append(self)
            return _orig(self)
        monkeypatch.setattr(getattr(pipeline, cls), "start", start)
    outs = {}
    for name, env in (("file", {}), ("callback", {"QP_NATIVE_FILE_SOURCE": "0"}), ("python", {"QP_NATIVE_PRODUCER": "0"})):
        for k in ("QP_NATIVE_FILE_SOURCE", "QP_NATIVE_PRODUCER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        obj = lvu.LVU(lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=4, num_frames=24), model=m)
        outs[name] = obj.generate("What is shown?", video, max_new_tokens=4)
        t = obj._pipeline.last_timings
        assert t.groups == 6 and t.producer_busy > 0 and t.gpu_prefill_busy > 0 and len(t.group_gaps) == 6, (name, t)
        assert t.gpu_stall_unknown == 0.0, (name, t)                          # every group's upload time is known (per-group events)
    assert outs["file"] == outs["callback"] == outs["python"] and outs["file"][0].count("<tok_") == 4
    kinds = [(type(p).__name__, getattr(p, "native_file", None)) for p in made]
    assert kinds == [("_NativeProducer", True), ("_NativeProducer", False), ("_Producer", None)], kinds
