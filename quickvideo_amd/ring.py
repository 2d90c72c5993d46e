"""Python face of the native frame ring (include/quickprefill.h: qp_frame_ring_*; csrc/qp_ring.hip).

The reference's overlap producer is a daemon Python thread that decodes, runs the HF processor under the GIL and feeds a Queue(3)
the main thread polls every 10 ms (lvu/models/qwen25_lvu_interleaved.py:237-342, 853-871).  Here the thread, the slot bookkeeping,
the H2D enqueue and the event hand-shakes live in the library; Python only supplies the frame source — a reader object whose next()
decodes straight into the pinned slot (`next_into`, frames.py), or, for pre-decoded .npy videos, nothing at all (the library
pread()s the frames itself).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np
import torch

from .native import FRAME_SOURCE_FN, QP_ERR_TIMEOUT, QP_OK, QuickPrefillError, host_memcpy, load_library

_LIB = None


def _lib():
    # the GIL-releasing handle: acquire() and stop() block on the producer thread, which may need the GIL for a Python source
    global _LIB
    if _LIB is None:
        _LIB = load_library(hold_gil=False)
    return _LIB


class FrameRing:
    """depth pinned host slots + depth device slots (the caller's tensors), one native producer thread per video.
    `ctx` = the qp_ctx handle of the device's QuickPrefillOps (None with dev_slots None: host-only ring, no GPU needed)."""

    POLL_MS = 100          # host wait per qp_frame_ring_acquire_for call

    def __init__(self, host_slots: List[torch.Tensor], dev_slots: Optional[List[torch.Tensor]] = None, ctx=None, copy_stream=None):
        self.lib = _lib()
        self.depth = len(host_slots)
        self.host, self.dev = host_slots, dev_slots
        self.frame_shape = tuple(host_slots[0].shape[1:])
        self.frame_bytes = int(np.prod(self.frame_shape))
        self.slot_bytes = host_slots[0].numel()
        assert all(t.dtype == torch.uint8 and t.is_contiguous() and t.numel() == self.slot_bytes for t in host_slots)
        assert (dev_slots is None) == (ctx is None), "a device ring needs the context and device slots; a host-only ring neither"
        arr = ctypes.c_void_p * self.depth
        hs = arr(*[t.data_ptr() for t in host_slots])
        ds = arr(*[t.data_ptr() for t in dev_slots]) if dev_slots is not None else None
        self._stream = copy_stream                                     # keeps the torch stream object alive as long as the ring
        h = ctypes.c_void_p()
        self._check(self.lib.qp_frame_ring_create(ctx, self.depth, self.slot_bytes, hs, ds,
                                                  copy_stream.cuda_stream if copy_stream is not None else None, ctypes.byref(h)))
        self.h = h
        self._cb = None
        self.exc: Optional[BaseException] = None
        self._origin = None

    def _check(self, rc):
        if rc != QP_OK:
            raise QuickPrefillError(rc, self.lib.qp_last_error().decode())

    # ---- sources
    def start_reader(self, reader, n_groups: int):
        """`reader`: any object with the InterleavedVideoReader contract.  Readers of this package decode into the slot
        (`next_into`); a foreign reader's next() result is copied into it by the native multi-threaded memcpy."""
        into = getattr(reader, "next_into", None)
        shape = self.frame_shape

        def source(_user, g, dst, capacity):
            try:
                n_max = capacity // self.frame_bytes
                view = np.ctypeslib.as_array(ctypes.cast(dst, ctypes.POINTER(ctypes.c_uint8)), shape=(n_max,) + shape)
                if into is not None:
                    return int(into(view)) * self.frame_bytes
                frames = next(reader)
                if not isinstance(frames, torch.Tensor):
                    frames = torch.from_numpy(np.ascontiguousarray(frames))
                frames = frames.contiguous()
                if tuple(frames.shape[1:]) != shape or frames.dtype != torch.uint8 or frames.shape[0] > n_max:
                    raise ValueError(f"reader returned {tuple(frames.shape)} {frames.dtype}; the ring holds groups of up to {n_max} uint8 frames of {shape}")
                host_memcpy(torch.from_numpy(view[: frames.shape[0]]), frames)
                return frames.shape[0] * self.frame_bytes
            except BaseException as e:      # re-raised in the consumer (qwen25_lvu_interleaved.py:291-292, 314-316 do the same with a queue item)
                self.exc = e
                return -1

        self._cb = FRAME_SOURCE_FN(source)
        self._check(self.lib.qp_frame_ring_start(self.h, ctypes.cast(self._cb, ctypes.c_void_p), None, n_groups))

    def start_file(self, path: str, data_offset: int, frame_idx, frames_per_group: int, io_threads: int = 8):
        idx = np.ascontiguousarray(frame_idx, dtype=np.int64)
        self._check(self.lib.qp_frame_ring_start_file(self.h, str(path).encode(), data_offset, self.frame_bytes,
                                                      idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(idx), frames_per_group, io_threads))

    # ---- consumer side
    def set_origin(self, event: "torch.cuda.Event"):
        self._origin = event                                           # alive as long as the ring
        self._check(self.lib.qp_frame_ring_set_origin(self.h, event.cuda_event))

    def acquire(self, g: int, consumer_stream=None) -> torch.Tensor:
        """Frames of group g: a view of the device slot (the consumer stream waits for the copy on the device), or of the host slot."""
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        cs = consumer_stream.cuda_stream if consumer_stream is not None else None
        # bounded waits in a loop: between two of them the interpreter runs its signal handlers, so Ctrl-C (or a consumer-side error
        # raised from another thread) ends a generate() whose frame source hangs — the reference's consumer polls its queue every
        # 10 ms for the same reason (qwen25_lvu_interleaved.py:853-871).  The wait itself is a condition variable: no latency added.
        while True:
            rc = self.lib.qp_frame_ring_acquire_for(self.h, g, cs, self.POLL_MS, ctypes.byref(ptr), ctypes.byref(nbytes))
            if rc != QP_ERR_TIMEOUT:
                break
        if rc != QP_OK:
            if self.exc is not None:
                raise self.exc
            self._check(rc)
        slot = (self.dev if self.dev is not None else self.host)[g % self.depth]
        assert ptr.value == slot.data_ptr()
        return slot[: nbytes.value // self.frame_bytes]

    def mark_read(self, g: int, consumer_stream=None):
        self._check(self.lib.qp_frame_ring_mark_read(self.h, g, consumer_stream.cuda_stream if consumer_stream is not None else None))

    def release(self, g: int, consumer_stream=None):
        self._check(self.lib.qp_frame_ring_release(self.h, g, consumer_stream.cuda_stream if consumer_stream is not None else None))

    def stop(self):
        if self.h:
            self.lib.qp_frame_ring_stop(self.h)

    def stats(self) -> dict:
        out = (ctypes.c_double * 6)()
        self._check(self.lib.qp_frame_ring_stats(self.h, out, 6))
        return {"busy": out[0], "wait_slot": out[1], "wait_h2d": out[2], "copy": out[3], "produced": int(out[4]), "groups": int(out[5])}

    def h2d_ms(self, n: int) -> List[float]:
        out = (ctypes.c_float * max(n, 1))()
        self._check(self.lib.qp_frame_ring_h2d_ms(self.h, out, n))
        return [float(out[i]) for i in range(n)]

    def close(self):
        if getattr(self, "h", None):
            self.lib.qp_frame_ring_destroy(self.h)
            self.h = None
        self._cb = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
