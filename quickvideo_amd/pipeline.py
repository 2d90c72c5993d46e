"""Video -> first token pipeline: CPU frame production overlapped with GPU ViT + group prefill.

Reference shape (lvu/models/qwen25_lvu_interleaved.py:237-342, 733-942): a daemon thread pulls frame groups from
the reader, runs the HF processor on the CPU and feeds a bounded Queue(3); the main thread polls it every 10 ms and
runs the group loop.  Here the producer is a native thread of the library (qp_frame_ring_*, csrc/qp_ring.hip): the frame source
decodes uint8 frames straight into a pinned host ring, the thread enqueues the async H2D copy on a dedicated HIP stream
(event-signalled in both directions, no polling); normalise + patchify + ViT run on the GPU on a second stream, one group ahead of the
LLM prefill that runs on the main stream.  Token ids and M-RoPE positions are
built before any pixel exists from (nframes, H, W) alone, like the reference's dummy_call (interleaved:522-638, 786-810).
"""
from __future__ import annotations

import os
import queue
import threading
import time
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import planner
from .engine import QuickPrefillEngine
from .decode import GraphDecoder
from .frames import open_video
from .native import host_memcpy
from .lvu_config import LVUConfig, effective_k
from .parallel import ParallelContext, stage_weights
from .processor import as_messages, prompt_from_messages
from .sampling import TokenSelector
from .spec import TextSpec
from .vit import VisionTower, VisionWeights, patchify_frames
from .weights import DecoderWeights


@dataclass
class QwenVLNative:
    """The `model` object of the native plugin: decoder + vision weights resident on one device."""
    text: DecoderWeights
    vision: VisionWeights
    device: torch.device
    name: str = "synthetic"
    rope_deltas: Optional[int] = None
    engine: Optional[QuickPrefillEngine] = None
    config: Optional[LVUConfig] = None
    generation_defaults: Optional[dict] = None      # the checkpoint's generation_config.json (HF generate applies it implicitly)
    parallel: Optional[ParallelContext] = None      # multi-GPU job this replica / shard belongs to (parallel.py); None = single process

    @property
    def spec(self) -> TextSpec:
        return self.text.spec


@dataclass
class Timings:
    """Seconds.  Host clocks: perf_counter; device intervals: HIP events of the three streams against one origin event.
    The reference reports `video_processing_time`, `total_prefill` and `e2e_time` from unsynchronised host clocks
    (qwen25_lvu_interleaved.py:852-905); here every figure says which resource it measures, so a reader can tell a slow
    producer from a busy GPU."""
    prefill: float = 0.0               # group loop wall time, device-synchronised at the end (total_prefill)
    decode: float = 0.0
    e2e: float = 0.0
    ttft: float = 0.0                  # video opened -> first token id on the host
    tokens: int = 0
    groups: int = 0
    # host side of the producer (the reference's decode + resize + processor thread)
    producer_busy: float = 0.0         # inside next(reader): decoding / resizing frames (the cost the overlap is meant to hide)
    producer_blocked: float = 0.0      # waiting for a free ring slot or for the previous H2D out of it: back-pressure, the GPU is the bottleneck
    producer_copy: float = 0.0         # filling the pinned slot + enqueueing the H2D copy
    sequential_fetch: float = 0.0      # overlap=False only: wall time of fetching EVERY group before the GPU starts (video_processing_time)
    consumer_get_wait: float = 0.0     # host thread blocked in queue.get() (it runs ahead of the GPU, so this is NOT GPU idle time)
    # device side (event timestamps)
    gpu_prefill_busy: float = 0.0      # sum over groups of prefill(g) on the main stream
    gpu_stall_frames: float = 0.0      # main stream idle between groups because group g's frames had not been uploaded yet: TRUE frame wait
    gpu_stall_vit: float = 0.0         # main stream idle because ViT(g) was not finished although its frames were there (ViT not hidden)
    gpu_stall_unknown: float = 0.0     # idle in front of a group whose upload time is not known (no stamp): booked to neither of the two
    vit_span: float = 0.0              # sum of ViT(g) start->end on its own stream WHILE the prefill shares the CUs (contended)
    vit_uncontended: float = 0.0       # ViT of one group replayed alone after the run, x groups: what the tower costs by itself
                                       # (only with PrefillPipeline.measure_vit_alone; 0 otherwise)
    group_gaps: Optional[list] = None  # per group: seconds the main stream sat idle in front of it (prefill(g-1) end -> prefill(g) start):
                                       # a late frame, a late ViT pass, or a launch thread that was not scheduled (host contention)
    layout: str = "single"             # multi-GPU grid this video ran on (parallel.py)


class _GpuProgress(threading.Thread):
    """QP_PIPELINE_DEBUG=1: every 5 s, how many groups the host has enqueued and how many ViT passes / group prefills the GPU has
    finished (event queries), on stderr — tells a slow device from a stuck one."""

    def __init__(self, n_groups):
        super().__init__(daemon=True)
        self.n, self.vit, self.pre, self.halt, self.t0 = n_groups, [], [], threading.Event(), time.perf_counter()
        self.start()

    def enqueued(self, vit_done, prefill_done):
        self.vit.append(vit_done); self.pre.append(prefill_done)

    def run(self):
        import sys
        while not self.halt.wait(5.0):
            nv, npf = sum(e.query() for e in list(self.vit)), sum(e.query() for e in list(self.pre))
            print(f"[pipeline {time.perf_counter() - self.t0:6.1f}s] enqueued {len(self.pre)}/{self.n} groups; finished on the GPU: "
                  f"{nv} ViT passes, {npf} group prefills", file=sys.stderr, flush=True)

    def stop(self):
        self.halt.set()


class _RemoteVideo:
    """What a rank other than 0 knows about the video: the four numbers rank 0 read from the container (frame count, frame rate,
    source size) — enough to derive the same plan (plan() needs no pixels); the frames themselves arrive per group by scatter."""

    def __init__(self, meta):
        self.total, self.fps, self.src_h, self.src_w, arr_hw = meta
        self.height = self.width = None
        if arr_hw is not None:                                       # array-backed source: frames are already at model size
            self.arr = np.empty((0, 3) + tuple(arr_hw), dtype=np.uint8)
            self.src_h = self.src_w = None

    def __len__(self):
        return self.total

    def get_fps(self):
        return self.fps


def _video_meta(reader):
    src_h = getattr(reader, "src_h", None) or reader.height
    src_w = getattr(reader, "src_w", None) or reader.width
    arr_hw = None if src_h is not None else tuple(int(v) for v in reader.arr.shape[2:])
    return (len(reader), float(reader.get_fps()), src_h, src_w, arr_hw)


class _Producer(threading.Thread):
    """Frame groups -> pinned ring -> async H2D on `copy_stream`; bounded like the reference's Queue(maxsize=3)."""

    def __init__(self, reader, n_groups, frames_per_group, device, depth=3, ring_cache=None, copy_stream=None):
        super().__init__(daemon=True)
        self.reader, self.n_groups, self.device, self.depth = reader, n_groups, device, depth
        # pinned host slots + device slots are kept across videos by the owner of `ring_cache` (page-locking ~80 MB costs tens
        # of ms: it would otherwise sit in front of the first group of every video)
        self.ring_cache = ring_cache if ring_cache is not None else {}
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self.exc = None
        self.use_gpu = device.type == "cuda"
        # a stream on a hardware queue of its own (streams.side_streams): a copy stream that shares a queue with the ViT or the LLM stream
        # uploads group g+1's frames only after that stream's earlier work has finished
        self.copy_stream = (copy_stream if copy_stream is not None else torch.cuda.Stream(device, priority=-1)) if self.use_gpu else None
        self.slots_free = threading.Semaphore(depth)
        self.ring = None
        self.fpg = frames_per_group
        self.h2d_done = [None] * depth      # per slot: event after the last H2D copy out of the pinned buffer (copy stream)
        self.read_done = [None] * depth     # per slot: event after the consumer's last GPU read of the device buffer (ViT stream)
        self.t_busy = self.t_blocked = self.t_copy = 0.0    # host seconds: in next(reader) / waiting for a slot / filling + enqueueing
        self.t_sem = self.t_sync = self.t_put = 0.0         # split of the waits: ring semaphore / previous H2D out of the pinned slot / full queue
        self.cancelled = threading.Event()

    def run(self):
        try:
            pc = time.perf_counter
            for g in range(self.n_groups):
                t0 = pc()
                frames = next(self.reader)                      # uint8 [g,3,H,W] (CPU work, GIL released inside numpy / PIL)
                t1 = pc()
                self.t_busy += t1 - t0
                self.slots_free.acquire()
                self.t_blocked += pc() - t1
                self.t_sem += pc() - t1
                if self.cancelled.is_set():                     # the consumer gave up (exception in the group loop): do not linger
                    return
                if self.use_gpu:
                    if self.ring is None:
                        shape = (self.fpg,) + tuple(frames.shape[1:])
                        key = (self.depth, shape, str(self.device))
                        if key not in self.ring_cache:
                            self.ring_cache.clear()                 # one video geometry at a time
                            self.ring_cache[key] = [(torch.empty(shape, dtype=torch.uint8).pin_memory(),
                                                     torch.empty(shape, dtype=torch.uint8, device=self.device)) for _ in range(self.depth)]
                        self.ring = self.ring_cache[key]
                    slot = g % self.depth
                    host, dev = self.ring[slot]
                    # slot reuse is ordered explicitly, not by luck: (1) the pinned host buffer may only be overwritten once the
                    # previous H2D copy out of it has finished (host-side wait, this thread only); (2) the device buffer may only
                    # be overwritten once the consumer's last GPU read of it (patchify on the ViT stream) has finished — the
                    # consumer hands that event back through release() and the copy stream waits on it.
                    t2 = pc()
                    if self.h2d_done[slot] is not None:
                        self.h2d_done[slot].synchronize()
                    t3 = pc()
                    self.t_blocked += t3 - t2
                    self.t_sync += t3 - t2
                    # native, GIL-free fill of the pinned slot (qp_host_memcpy through ctypes; 2-10x faster than Tensor.copy_ here,
                    # whose speed follows torch's process-wide intra-op thread count): the launching thread is never starved
                    host_memcpy(host[: frames.shape[0]], frames.contiguous())
                    with torch.cuda.stream(self.copy_stream):
                        if self.read_done[slot] is not None:
                            self.copy_stream.wait_event(self.read_done[slot])
                        dev[: frames.shape[0]].copy_(host[: frames.shape[0]], non_blocking=True)
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(self.copy_stream)
                    self.h2d_done[slot] = ev
                    t4 = pc()
                    self.t_copy += t4 - t3
                    self.q.put((g, dev[: frames.shape[0]], ev))
                    self.t_put += pc() - t4
                else:
                    self.q.put((g, frames.clone(), None))
        except BaseException as e:   # re-raised in the consumer, like interleaved:291-292, 314-316
            self.exc = e
            self.q.put(None)

    def get(self):
        item = self.q.get()
        if item is None:
            raise self.exc
        return item

    def cancel(self):
        """Called by the consumer when it leaves the group loop early: wakes the thread wherever it waits and lets it end."""
        self.cancelled.set()
        for _ in range(self.depth + 1):
            self.slots_free.release()
        try:
            while True:
                self.q.get_nowait()                               # a put() blocked on the full queue returns
        except queue.Empty:
            pass

    def release(self, g: int = 0, read_done=None):
        """Group g's slot may be refilled; `read_done` = event recorded after the last GPU read of its device buffer."""
        self.read_done[g % self.depth] = read_done
        self.slots_free.release()

    def mark_read(self, g: int):
        pass                                                          # the consumer's torch event, handed to release(), plays this role

    def h2d_ms(self, origin):
        return None                                                   # the consumer holds the torch events of the copies itself


class _NativeProducer:
    """The same seat as `_Producer`, filled by the library: the thread, the slot bookkeeping, the H2D enqueue and both event
    hand-shakes are qp_frame_ring_* (csrc/qp_ring.hip; ring.py) — the product path on a GPU.  A reader of this package decodes
    straight into the pinned slot (no intermediate array, no memcpy); a pre-decoded .npy video is read by the library itself and
    never touches the interpreter; any other reader's next() result is copied in by the native memcpy.
    get() makes `consumer_stream` wait for the copy on the device, so the consumer gets no event to wait on (ev = None)."""

    def __init__(self, reader, n_groups, frames_per_group, frame_hw, device, ctx, consumer_stream, depth=3, ring_cache=None, copy_stream=None):
        from .ring import FrameRing
        self.reader, self.n_groups, self.depth, self.device = reader, n_groups, depth, device
        self.consumer_stream = consumer_stream
        self.copy_stream = copy_stream if copy_stream is not None else torch.cuda.Stream(device, priority=-1)
        shape = (frames_per_group, 3, int(frame_hw[0]), int(frame_hw[1]))   # the plan's frame size (the reader may be a bare iterator)
        self.ring_cache = ring_cache if ring_cache is not None else {}
        key = (depth, shape, str(device))
        if key not in self.ring_cache:
            self.ring_cache.clear()                                   # one video geometry at a time
            self.ring_cache[key] = [(torch.empty(shape, dtype=torch.uint8).pin_memory(),
                                     torch.empty(shape, dtype=torch.uint8, device=device)) for _ in range(depth)]
        slots = self.ring_cache[key]
        self.ring = FrameRing([h for h, _ in slots], [d for _, d in slots], ctx=ctx, copy_stream=self.copy_stream)
        self.fpg = frames_per_group
        self.native_file = False
        self._next = 0                                                # next group the consumer will ask for (get() is called in order)
        self.t_busy = self.t_blocked = self.t_copy = self.t_sem = self.t_sync = self.t_put = 0.0

    def start(self):
        layout = getattr(self.reader, "raw_layout", None)
        layout = layout() if callable(layout) else None
        pending = getattr(self.reader, "pending_indices", None)
        pending = pending() if callable(pending) else None
        ok = layout is not None and pending is not None and os.environ.get("QP_NATIVE_FILE_SOURCE", "1") != "0"
        if ok:
            path, off, frame_bytes = layout
            stored = getattr(self.reader, "stored_hw", None)
            # the library pread()s ring.frame_bytes per frame: only when that IS the file's frame (the plan's H x W equals the stored
            # size); anything else goes through the reader, whose own check raises the ValueError the user should see
            ok = int(frame_bytes) == self.ring.frame_bytes and (not callable(stored) or tuple(stored()) == tuple(self.ring.frame_shape[1:]))
        if ok:
            self.native_file = True
            self.ring.start_file(path, off, pending, self.fpg, io_threads=max(1, int(getattr(self.reader, "num_threads", 8))))
            self.reader.advance(len(pending))                          # the ring now owns the rest of the selection
        else:
            self.ring.start_reader(self.reader, self.n_groups)

    def set_origin(self, event):
        self.ring.set_origin(event)

    def get(self):
        g = self._next
        self._next += 1
        return g, self.ring.acquire(g, self.consumer_stream), None

    def mark_read(self, g: int):
        self.ring.mark_read(g, self.consumer_stream)

    def release(self, g: int = 0, read_done=None):
        self.ring.release(g, self.consumer_stream)

    def cancel(self):
        self.ring.stop()

    def finish(self):
        """After the last group was acquired and released: totals of the producer thread, then the ring is closed."""
        st = self.ring.stats()
        self.t_busy, self.t_copy = st["busy"], st["copy"]
        self.t_sem, self.t_sync = st["wait_slot"], st["wait_h2d"]
        self.t_blocked = self.t_sem + self.t_sync

    def h2d_ms(self, origin):
        return self.ring.h2d_ms(self.n_groups)

    def close(self):
        self.ring.close()


_WARNED: set = set()


def _warn_once(key: str, msg: str):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, stacklevel=3)


_SWITCH_LOCK = threading.Lock()
_SWITCH_STATE = {"depth": 0, "saved": None}


class _switch_interval_override:
    """sys.setswitchinterval is process-wide: the override is ref-counted — the first generate() entering saves the interval and
    lowers it, the last one leaving restores the saved value (two overlapping generates used to restore 5 ms while the other was still
    running, or save the lowered value as "old" and leave it set for good).  want <= 0: no override."""

    def __init__(self, want: float):
        self.want = want

    def __enter__(self):
        import sys
        if self.want <= 0:
            return self
        with _SWITCH_LOCK:
            if _SWITCH_STATE["depth"] == 0:
                _SWITCH_STATE["saved"] = sys.getswitchinterval()
                sys.setswitchinterval(min(_SWITCH_STATE["saved"], self.want))
            _SWITCH_STATE["depth"] += 1
        return self

    def __exit__(self, *exc):
        import sys
        if self.want <= 0:
            return False
        with _SWITCH_LOCK:
            _SWITCH_STATE["depth"] -= 1
            if _SWITCH_STATE["depth"] == 0 and _SWITCH_STATE["saved"] is not None:
                sys.setswitchinterval(_SWITCH_STATE["saved"])
                _SWITCH_STATE["saved"] = None
        return False


class PrefillPipeline:
    def __init__(self, model: QwenVLNative, config: LVUConfig, processor, ops=None):
        self.model, self.cfg, self.processor, self.ops = model, config, processor, ops
        self.use_gpu = model.device.type == "cuda"
        self.par: ParallelContext = getattr(model, "parallel", None) or ParallelContext()
        self._vit_pg = None                       # dedicated process group (own RCCL communicator) of the front end's collectives
        self.last_layout = "single"
        self._tower = None
        self.vit_stream = self.copy_stream = None
        self.stream_report = None
        if self.use_gpu:
            from .streams import side_streams
            self.vit_stream, self.copy_stream, self.stream_report = side_streams(model.device)
        self.last_timings: Optional[Timings] = None
        self.measure_vit_alone = False            # bench.py: replay one group's ViT pass after the run, GPU otherwise idle (3 extra passes)

    @property
    def tower(self) -> VisionTower:
        if self._tower is None:
            ops = self.ops
            if ops is None and self.use_gpu:
                from .native import QuickPrefillOps
                ops = self.ops = QuickPrefillOps(self.model.device)
            self._tower = VisionTower(self.model.vision, ops=ops if self.use_gpu else None)
        return self._tower

    # ------------------------------------------------------------------ planning (no pixels needed)
    @staticmethod
    def video_entry(question):
        """The single video entry of a `messages` list (qwen25_lvu.py:552-554: `extract_vision_info`, one video only); None for a bare
        question."""
        if isinstance(question, str):
            return None
        entries = [c for m in question if not isinstance(m["content"], str) for c in m["content"] if c.get("type") == "video" or "video" in c]
        assert len(entries) == 1, "Only one video is supported for now."
        return entries[0]

    def plan(self, reader, question):
        """`question`: the reference's `messages` list (one video entry; qwen25_lvu.py:546-554) or a bare question.  Frame count and frame
        size come from the VIDEO ENTRY of the message — fps / nframes / min_frames / max_frames / min_pixels / max_pixels / total_pixels /
        resized_height / resized_width, exactly the keys the reference's fetch_video + smart_nframes read (qwen25_lvu.py:351-370, 402-442;
        interleaved:343-383, 416-436) with qwen-vl-utils' defaults (fps 2.0) for what the entry leaves out.  A bare question gets the
        entry run_lvu_model builds from the LVUConfig (fps xor num_frames, extra_kwargs max/min_pixels; :504-536).  Pinned by GV4."""
        cfg, spec = self.cfg, self.model.spec
        total, vfps = len(reader), reader.get_fps()
        ele = self.video_entry(question)
        if ele is None:
            ele = planner.video_entry_from_config(None, cfg.fps, cfg.num_frames, cfg.extra_kwargs)
        if "video_start" in ele or "video_end" in ele:
            raise NotImplementedError("not support start_pts and end_pts in deepcodec for now.")        # interleaved:396-397
        nframes = planner.smart_nframes(ele, total, vfps)
        src_h = getattr(reader, "src_h", None) or reader.height
        src_w = getattr(reader, "src_w", None) or reader.width
        if src_h is None:                                            # array-backed video: frames already at model size
            src_h, src_w = reader.arr.shape[2:]
            H, W = src_h, src_w
        else:
            H, W = planner.video_frame_size(nframes, src_h, src_w, ele)
        idx = np.linspace(0, total - 1, nframes).round().astype(np.int64)   # interleaved:397-399
        vs = self.model.vision.spec
        gh, gw = H // vs.patch_size, W // vs.patch_size
        # the processor seam (lvu/lvu.py:18-23): the caller's own processor renders the chat template and tokenises it
        prompt = prompt_from_messages(self.processor, as_messages(question))
        n_video = (nframes // vs.temporal_patch_size) * (gh // 2) * (gw // 2)
        T = len(prompt.prefix_ids) + n_video + len(prompt.tail_ids)
        gs = cfg.video_group_size
        plan = planner.plan_groups(nframes, gs, gh, gw, len(prompt.prefix_ids), T, vs.temporal_patch_size, vs.spatial_merge_size)
        # Qwen2.5-VL: tokens_per_second * second_per_grid_t with second_per_grid_t = temporal_patch / sampled fps
        sample_fps = nframes / max(total, 1e-6) * vfps             # the reference's own expression and operation order (qwen25_lvu.py:243, interleaved:408)
        q25 = spec.temporal_scale < 0                                # Qwen2.5-VL: HF's float32 expression (planner.temporal_ids)
        pos, delta = planner.mrope_positions(len(prompt.prefix_ids), (nframes // vs.temporal_patch_size, gh, gw), len(prompt.tail_ids),
                                             vs.spatial_merge_size, spec.temporal_scale if not q25 else 1.0,
                                             second_per_grid_t=vs.temporal_patch_size / sample_fps if q25 else None,
                                             tokens_per_second=-spec.temporal_scale if q25 else None)
        return dict(nframes=nframes, H=H, W=W, idx=idx, prompt=prompt, plan=plan, pos=pos, delta=delta, T=T, gh=gh, gw=gw)

    def _engine(self, plan, T, max_new_tokens: int = 0) -> QuickPrefillEngine:
        cfg, spec, par = self.cfg, self.model.spec, self.par
        kept = sum((effective_k(n, cfg, 0, spec.n_layers) or n) for n in plan.tokens)
        need_cap = kept + plan.tail_len + max(256, max_new_tokens + 8)        # room for every token that will be decoded
        n_max = max(plan.tokens + [plan.tail_len, 1])
        if cfg.query_based:                                   # the prompt tokens ride along with every group (qwen25_lvu.py:684-686)
            n_max += plan.tail_len
        # multi-GPU job: the pp x sp grid is a per-VIDEO decision (parallel.py: long videos pipeline their layers, short ones split the
        # group's rows); tensor parallelism is fixed when the weights are loaded (they are sharded)
        pp, sp = par.grid(len(plan.tokens), spec.n_layers)
        self.last_layout = par.describe(pp, sp)
        eng = self.model.engine
        if (eng is None or eng.arena.capacity < need_cap or eng.n_max < n_max or eng.cfg is not cfg or getattr(eng, "_grid", (1, 1)) != (pp, sp)):
            self.model.engine = eng = None                                    # free the old arena before the new one is allocated
            w = stage_weights(self.model.text, pp, par.rank // sp) if par.on else self.model.text
            eng = QuickPrefillEngine(w, cfg, capacity=need_cap, max_group_tokens=n_max, device=self.model.device, ops=self.ops,
                                     **par.engine_kwargs(pp, sp))
            eng._grid = (pp, sp)
            self.model.engine = eng
        eng.reset()
        return eng

    # ------------------------------------------------------------------ multi-GPU front end ("ViT: data-parallel over frames")
    def _front_group(self):
        """The front end's collectives (frame scatter, feature all-gather, token broadcast) run on their OWN process group = their own
        RCCL communicator and stream: on the job's main communicator they would queue behind the layer pipeline's point-to-point
        hand-offs (torch shares one communicator between collectives and send/recv when it was created eagerly), and the all-gather of
        group g+1's features would then wait for every stage to finish group g — serialising the pipe."""
        if self._vit_pg is None:
            dist = torch.distributed
            from .parallel import dist_timeout
            self._vit_pg = dist.new_group(ranks=[self.par.global_rank(r) for r in range(self.par.world)], timeout=dist_timeout())
        return self._vit_pg

    def _vit_parallel(self, frames, n_frames: int, H: int, W: int):
        """ViT of one frame group, data-parallel over its frame pairs: rank 0 holds the uint8 frames [n_frames, 3, H, W]; they are
        SCATTERED by temporal patch (frame pair) in contiguous chunks of ceil(pairs / world), every rank runs normalise + patchify +
        the tower on its chunk — a temporal patch is its own attention sequence in both Qwen towers, so the split changes no
        arithmetic — and ONE all-gather assembles the [n, d] features in token order on every rank.
        -> (features [n, d], event after the last read of `frames` or None)."""
        par, dist, dev = self.par, torch.distributed, self.model.device
        vs, grp = self.model.vision.spec, self._front_group()
        tp = vs.temporal_patch_size
        pairs = n_frames // tp
        chunk = -(-pairs // par.world)
        S = (H // vs.patch_size // vs.spatial_merge_size) * (W // vs.patch_size // vs.spatial_merge_size)   # tokens per temporal patch
        mine = max(0, min(chunk, pairs - par.rank * chunk))
        dt = self.model.text.embed.dtype
        recv = torch.empty(chunk * tp, 3, H, W, dtype=torch.uint8, device=dev)
        src = par.global_rank(0)
        gloo_on_gpu = self.use_gpu and dist.get_backend(grp) == "gloo"          # developer runs of several ranks on ONE GPU
        if gloo_on_gpu:                                                        # gloo moves CUDA tensors by broadcast / all-reduce only
            whole = frames.contiguous() if par.rank == 0 else torch.empty(n_frames, 3, H, W, dtype=torch.uint8, device=dev)
            dist.broadcast(whole, src=src, group=grp)
            recv[: mine * tp].copy_(whole[par.rank * chunk * tp: (par.rank * chunk + mine) * tp])
        else:
            parts = None
            if par.rank == 0:
                if pairs == chunk * par.world:
                    padded = frames.contiguous()
                else:                                                          # ragged last group: pad to world equal chunks
                    padded = torch.zeros(par.world * chunk * tp, 3, H, W, dtype=torch.uint8, device=dev)
                    padded[:n_frames].copy_(frames)
                parts = list(padded.view(par.world, chunk * tp, 3, H, W).unbind(0))
            dist.scatter(recv, parts, src=src, group=grp)
        read_done = None
        if self.use_gpu:
            read_done = torch.cuda.Event()
            read_done.record(torch.cuda.current_stream(dev))                    # rank 0: the ring's device slot has been read
        local = torch.zeros(chunk * S, self.model.spec.hidden, dtype=dt, device=dev)
        if mine:
            rows, grid = self.tower.patchify(recv[: mine * tp])
            local[: mine * S].copy_(self.tower.forward(rows, grid))
        allf = torch.empty(par.world * chunk * S, self.model.spec.hidden, dtype=dt, device=dev)
        dist.all_gather_into_tensor(allf, local, group=grp)
        if pairs != chunk * par.world:                                          # drop the padding rows of the last ranks
            allf = allf[: pairs * S]                                            # (chunks are contiguous and only the LAST ranks are short)
        return allf, read_done

    def _share_token(self, tok, src_rank: int) -> int:
        """The token the designated rank selected -> every rank (one int64 broadcast; under the layer pipeline only the last stage holds
        logits, and in every layout one rank decides so that ranks cannot drift apart on a rounding difference)."""
        dist = torch.distributed
        t = torch.tensor([int(tok) if tok is not None else 0], dtype=torch.int64, device=self.model.device)
        dist.broadcast(t, src=self.par.global_rank(src_rank), group=self._front_group())
        return int(t.item())

    # ------------------------------------------------------------------ video -> tokens
    def generate(self, *args, **kwargs) -> List[int]:
        """video -> tokens (see _generate).  While the group loop runs, the interpreter's thread switch interval is lowered to 20 us:
        every torch call of the launching thread (a few hundred per frame group) drops the interpreter lock, and a Python thread that is
        busy beside it — the reference's own processor thread is one, qwen25_lvu_interleaved.py:303-340 — then keeps the lock for a whole
        switch interval (5 ms by default) before the launcher gets it back: measured on cfg4s with ONE such thread, 9x slower
        (bench.py host_contention).  The other thread loses nothing but a little switching overhead; the setting is restored when the LAST concurrent generate of the
        process returns (ref-counted: two LVU objects may generate at the same time)."""
        with _switch_interval_override(float(os.environ.get("QP_SWITCH_INTERVAL_S", "2e-5"))):
            return self._generate(*args, **kwargs)

    @torch.no_grad()
    def _generate(self, question, video, max_new_tokens: int = 16, overlap: bool = True, eos_token_id=None,
                  do_sample: Optional[bool] = None, temperature: Optional[float] = None, top_k: Optional[int] = None,
                  top_p: Optional[float] = None, repetition_penalty: Optional[float] = None, seed: Optional[int] = None,
                  num_beams: int = 1, length_penalty: float = 1.0, early_stopping=False, **unused) -> List[int]:
        """generation kwargs as the reference hands them to HF `generate` (qwen25_lvu.py:744-761); unset ones fall back to the
        checkpoint's generation_config.json (`model.generation_defaults`), then to greedy.  `num_beams > 1`: beam search with HF's
        semantics (`length_penalty`, `early_stopping`, with `do_sample` beam-search multinomial sampling; beam.py).
        `question`: the user's text, or the reference's `messages` list (chat(); qwen25_lvu.py:546-548) when the processor can
        template it.  eos_token_id: an id or a list of ids (HF stops on ANY of generation_config.eos_token_id)."""
        num_beams = int(num_beams)
        gd = getattr(self.model, "generation_defaults", None) or {}
        pick = lambda v, k: gd.get(k) if v is None else v
        selector = TokenSelector(pick(do_sample, "do_sample") or False, pick(temperature, "temperature"), pick(top_k, "top_k"),
                                 pick(top_p, "top_p"), pick(repetition_penalty, "repetition_penalty"), seed, device=self.model.device)
        eos = pick(eos_token_id, "eos_token_id")
        eos_set = frozenset() if eos is None else frozenset(int(e) for e in (eos if isinstance(eos, (list, tuple, set, frozenset)) else [eos]))
        tm = Timings()
        dev = self.model.device
        par = self.par
        lead = par.rank == 0                                          # rank 0 owns the frame source and the producer thread
        t_e2e = time.perf_counter()
        if par.on:
            # every rank derives the same plan from (frame count, fps, source size): rank 0 reads them from the container and shares them
            box = [None]
            if lead:
                reader = open_video(video)
                box[0] = _video_meta(reader)
            obj_dev = dev if (self.use_gpu and torch.distributed.get_backend(self._front_group()) != "gloo") else None
            torch.distributed.broadcast_object_list(box, src=par.global_rank(0), group=self._front_group(), device=obj_dev)
            if not lead:
                reader = _RemoteVideo(box[0])
        else:
            reader = open_video(video)
        P = self.plan(reader, question)
        plan, pos = P["plan"], torch.from_numpy(P["pos"]).to(dev)
        gs = plan.frames[0]
        if lead:
            reader.height, reader.width, reader.interpolation = P["H"], P["W"], "LANCZOS"
            reader.frame_iter = gs
            reader.process(P["idx"])                                  # decoding starts here (interleaved:438-442)
        eng = self._engine(plan, P["T"], max_new_tokens)
        self.model.rope_deltas = P["delta"]                           # qwen25_lvu.py:620
        prefix = torch.tensor(P["prompt"].prefix_ids, dtype=torch.long, device=dev)
        tail = torch.tensor(P["prompt"].tail_ids, dtype=torch.long, device=dev)
        G = len(plan.tokens)
        if not overlap and lead:
            # sequential plugin (the reference's non-interleaved path, qwen25_lvu.py:551-575): EVERY frame group is fetched before the
            # GPU sees the first one; the groups then go through the same 3-slot ring (a memcpy + H2D each) and the same enqueue-ahead
            # ViT as the overlapped mode, so the GPU side of the two modes is identical and ttft(sequential) - ttft(overlapped) is what
            # the overlap hides
            t0 = time.perf_counter()
            fetched = [next(reader) for _ in range(G)]
            tm.sequential_fetch = time.perf_counter() - t0
            reader = iter(fetched)
        if not hasattr(self, "_ring_cache"):
            self._ring_cache = {}
        prod = None
        if lead:
            # bounded like the reference's Queue(maxsize=3).  On a GPU the producer is the library's native thread (qp_frame_ring_*);
            # the Python thread remains for the CPU test double and as an A/B (QP_NATIVE_PRODUCER=0)
            ctx = None
            if self.use_gpu and os.environ.get("QP_NATIVE_PRODUCER", "1") != "0":
                _ = self.tower
                ctx = getattr(self.ops, "ctx", None)
            if ctx is not None:
                prod = _NativeProducer(reader, G, gs, (P["H"], P["W"]), dev, ctx, self.vit_stream, depth=3, ring_cache=self._ring_cache,
                                       copy_stream=self.copy_stream)
            else:
                prod = _Producer(reader, G, gs, dev, depth=3, ring_cache=self._ring_cache, copy_stream=self.copy_stream)
            prod.start()
        sync = (lambda: torch.cuda.synchronize(dev)) if self.use_gpu else (lambda: None)
        ev_t = (lambda: torch.cuda.Event(enable_timing=True)) if self.use_gpu else (lambda: None)

        def vit_group(g, gate=None):
            """`gate`: an event on the main stream that ViT(g) must not start before — the start of prefill(g-2): the tower runs at most
            TWO groups ahead of the LLM, in both plugins (bounded feature buffers; and a sequential plugin whose frames are all there
            gets the same GPU schedule as the overlapped one).  Not one group: ViT(g+1) confined to the window of prefill(g) leaves
            the LLM stream idle whenever the contended tower needs longer than that one prefill (short groups: cfg4s lost 9 %).
            (Round 3's 1.5 s difference between the two plugins had another cause — the copy stream shared a hardware queue with the
            ViT / LLM stream, streams.py.)"""
            frames = ev = None
            if lead:
                t0 = time.perf_counter()
                gi, frames, ev = prod.get()
                tm.consumer_get_wait += time.perf_counter() - t0
            if self.use_gpu:
                s_ev, e_ev = ev_t(), ev_t()
                if ev is not None:
                    self.vit_stream.wait_event(ev)
                if gate is not None:
                    self.vit_stream.wait_event(gate)
                with torch.cuda.stream(self.vit_stream):
                    s_ev.record(self.vit_stream)
                    if par.on:                                    # scatter by frame pair -> ViT on this rank's share -> all-gather
                        feats, read_done = self._vit_parallel(frames, plan.frames[g], P["H"], P["W"])
                    else:
                        rows, grid = self.tower.patchify(frames)
                        read_done = torch.cuda.Event()
                        read_done.record(self.vit_stream)         # last read of the ring's device slot
                        prod.mark_read(g)                         # (native ring: its own event, recorded at the same point)
                        feats = self.tower.forward(rows, grid)
                    e_ev.record(self.vit_stream)
                return feats, (s_ev, e_ev, ev), read_done, frames
            if par.on:
                return self._vit_parallel(frames, plan.frames[g], P["H"], P["W"])[0], None, None, frames
            rows, grid = self.tower.patchify(frames)
            return self.tower.forward(rows, grid), None, None, frames

        t_pre = time.perf_counter()
        origin = ev_t()
        if self.use_gpu:
            origin.record(torch.cuda.current_stream(dev))
            if isinstance(prod, _NativeProducer):
                prod.set_origin(origin)
        start, trace = 0, []                                          # trace[g] = (h2d_done, vit_start, vit_end, prefill_start, prefill_end)
        dbg = _GpuProgress(G) if (self.use_gpu and os.environ.get("QP_PIPELINE_DEBUG")) else None
        # query-based predict types: the prompt (everything after the last video token) is appended to every group and scores its
        # keys (qwen25_lvu.py:661-664, 684-689); positions are then the group's AND the next tail_len of the sequence
        q_m = plan.tail_len if (self.cfg.query_based and self.cfg.enable) else 0
        tail_emb = eng.embed_tokens(tail) if q_m else None
        last_frames = None
        p0_prev = None                                                # start of the previous group's prefill (the ViT's run-ahead gate)
        ahead = (getattr(eng, "pp_size", 1) or 1) + 3
        bar = None
        if self.cfg.use_tqdm and lead:                                # the reference wraps its group loop in tqdm when asked (qwen25_lvu.py:671-672)
            try:
                from tqdm import tqdm
                bar = tqdm(total=G, desc="Processing video groups", disable=False)
            except ImportError:                                       # no tqdm in the environment: say so once, keep going
                _warn_once("use_tqdm", "LVUConfig.use_tqdm=True but tqdm is not importable: no progress bar")
        try:
            nxt = vit_group(0)
            for g, n in enumerate(plan.tokens):
                if bar is not None:
                    bar.update(1)                                     # groups ENQUEUED (the GPU runs behind the host by design)
                feats, evs, read_done, last_frames = nxt
                if self.use_gpu:
                    main = torch.cuda.current_stream(dev)
                    main.wait_event(evs[1])
                    # feats was allocated on the ViT stream and is read on the main stream (cat / copy into the engine's buffer): tell
                    # the caching allocator, or ViT(g+2) could be handed the same block while prefill(g) is still queued
                    feats.record_stream(main)
                    p0 = ev_t(); p0.record(main)
                emb = torch.cat([eng.embed_tokens(prefix), feats], 0) if g == 0 else feats
                assert emb.shape[0] == n, (emb.shape, n)
                # prefill(g) is enqueued BEFORE the host asks for group g+1's frames: with a producer-bound source (short videos)
                # prod.get() blocks until they exist, and the GPU must not sit idle behind that wait (round 2 had the two the other
                # way round: cfg2's overlapped TTFT carried ~70 ms of it).  ViT(g+1) still runs ahead on its own stream.
                eng.prefill_group(emb, pos[:, start:start + n + q_m], prompt_embeds=tail_emb)
                if self.use_gpu:
                    p1 = ev_t(); p1.record(torch.cuda.current_stream(dev))
                    trace.append((evs[2], evs[0], evs[1], p0, p1))
                    # a rank without the producer has nothing that holds its host thread back: keep it at most a pipeline's depth
                    # (+ the ring's 3 groups) ahead of its own GPU, so that queued feature buffers stay bounded
                    if par.on and not lead and len(trace) > ahead:
                        trace[-ahead - 1][4].synchronize()
                if lead:
                    prod.release(g, read_done)
                if g + 1 < G:
                    # ViT of the next group: own stream, beside the prefill.  The run-ahead gate is rank 0's alone: its scatter feeds every
                    # rank's share of the tower, so gating it there bounds them all — while a gate on a LATE pipeline stage's own prefill
                    # (pp-1 groups behind stage 0) would hold back the all-gather stage 0 is waiting for and cap a pipe of more than
                    # three stages at 2/(pp-1) of its rate
                    nxt = vit_group(g + 1, gate=p0_prev if (self.use_gpu and lead) else None)
                    if self.use_gpu:
                        p0_prev = p0
                start += n
                if dbg is not None:
                    dbg.enqueued(evs[1], p1)
        except BaseException:            # leave no producer thread behind that waits for a slot nobody will release
            if prod is not None:
                prod.cancel()
                if isinstance(prod, _NativeProducer):
                    prod.close()
            raise
        finally:
            if bar is not None:
                bar.close()
        sync()
        if dbg is not None:
            dbg.stop()
        tm.prefill = time.perf_counter() - t_pre
        if isinstance(prod, _NativeProducer):                         # every group acquired and released: totals, copy timestamps, ring closed
            prod.cancel()
            prod.finish()
            ms = prod.h2d_ms(origin)
            trace = [(ms[i],) + t[1:] for i, t in enumerate(trace)]
            prod.close()
        tm.tokens, tm.groups = start, G
        t_dec = time.perf_counter()
        logits = eng.prefill_tail(eng.embed_tokens(tail), pos[:, start:])     # pruning off for the tail (qwen25_lvu.py:737-742)
        if not selector.trivial:
            # HF's processors see `input_ids` of the generate() call.  The reference calls it with the TAIL only — everything behind
            # the last video token, `whole_inputs['input_ids'][:, past_len:]` over a pre-filled cache (qwen25_lvu.py:724-740) — so the
            # repetition penalty touches tail + generated tokens, never the system prompt or the video pads.
            selector.observe(list(P["prompt"].tail_ids), logits.shape[-1], dev)
        # who decides the token: the rank that holds the logits — the last pipeline stage's first rank; rank 0 otherwise (tensor /
        # group-token parallel ranks all hold them) — and every other rank receives its choice
        decider = (eng.pp_size - 1) * getattr(eng, "sp_size", 1) if (par.on and eng.pp_size > 1) else 0

        def choose(lg):
            t = None
            if not par.on or par.rank == decider:
                t = selector.select(lg) if not selector.trivial else int(torch.argmax(lg).item())
            return self._share_token(t, decider) if par.on else t

        if num_beams > 1:
            # beam search (HF generate semantics, beam.py): all beams share the prefilled video + prompt rows of the arena; the clock for
            # "first token" stops when the first step's distribution is on the host (the final first token is only known at the end)
            from .beam import EngineBeams, beam_search
            first = logits.float() if logits is not None else None
            if first is not None:
                _ = float(first[0].item())
            tm.ttft = time.perf_counter() - t_e2e
            beams = EngineBeams(eng, P["delta"], num_beams, max_new_tokens)
            search = lambda adv: beam_search(first, adv, num_beams, max_new_tokens, eos_ids=sorted(eos_set), length_penalty=float(length_penalty),   # noqa: E731
                                             early_stopping=early_stopping, repetition_penalty=selector.penalty, prompt_ids=list(P["prompt"].tail_ids),
                                             do_sample=selector.do_sample, temperature=selector.temperature, top_k=selector.top_k,
                                             top_p=selector.top_p, generator=selector.gen)
            if not par.on:
                out = search(beams.advance)
            else:
                # multi-GPU: ONE rank (the one that holds logits) runs the search; before every step it tells the others which beams to extend
                # by which tokens, and every rank advances its share of the model (its layers / its heads) in lock-step
                dist, grp, B = torch.distributed, self._front_group(), num_beams
                src = par.global_rank(decider)
                msg = torch.zeros(1 + 2 * B + max_new_tokens + 1, dtype=torch.int64, device=dev)

                def leader_advance(parents, tokens):
                    msg.zero_()
                    msg[0] = 1
                    msg[1:1 + B] = torch.tensor(parents, device=dev); msg[1 + B:1 + 2 * B] = torch.tensor(tokens, device=dev)
                    dist.broadcast(msg, src=src, group=grp)
                    return beams.advance(parents, tokens)

                if par.rank == decider:
                    out = search(leader_advance)
                    msg.zero_()
                    msg[1 + 2 * B] = len(out)
                    msg[2 + 2 * B:2 + 2 * B + len(out)] = torch.tensor(out, device=dev)
                    dist.broadcast(msg, src=src, group=grp)
                else:
                    while True:
                        dist.broadcast(msg, src=src, group=grp)
                        m_ = msg.tolist()
                        if m_[0] == 0:
                            out = m_[2 + 2 * B:2 + 2 * B + m_[1 + 2 * B]]
                            break
                        beams.advance(m_[1:1 + B], m_[1 + B:1 + 2 * B])
            beams.finish(len(out))
            eng.pp_flush()
            sync()
            tm.decode = time.perf_counter() - t_dec
            tm.e2e = time.perf_counter() - t_e2e
            if prod is not None:
                tm.producer_busy, tm.producer_blocked, tm.producer_copy = prod.t_busy, prod.t_blocked + prod.t_put, prod.t_copy
            if self.use_gpu and trace and lead:
                self._device_breakdown(tm, origin, trace)
            tm.layout = self.last_layout
            self.last_timings = tm
            return out
        tok = choose(logits)                                                  # first token on the host = TTFT point
        tm.ttft = time.perf_counter() - t_e2e
        out = [tok]
        graph = (not par.on) and GraphDecoder.supported(eng)                  # one hipGraph replay per token (decode.py); multi-GPU: eager
        if graph and getattr(eng, "_graph_decoder", None) is None:
            eng._graph_decoder = GraphDecoder(eng)
        if graph and selector.trivial:                                        # argmax stays on the device inside the graph
            out += eng._graph_decoder.generate(tok, max_new_tokens - 1, P["delta"], eos_set or None)
        else:                                                                 # logits come back per step: processors / sampling, or the
            if graph and max_new_tokens > 1:                                  # per-op path (CPU test doubles, parallel engines)
                eng._graph_decoder.begin(P["delta"])
            for _ in range(max_new_tokens - 1):
                if tok in eos_set:
                    break
                if graph:
                    logits = eng._graph_decoder.step(tok)
                else:
                    logits = eng.decode_step(eng.embed_tokens(torch.tensor([tok], device=dev)), P["delta"])
                tok = choose(logits)
                out.append(tok)
        eng.pp_flush()                                                        # layer pipeline: no hand-off left in flight
        sync()
        tm.decode = time.perf_counter() - t_dec
        tm.e2e = time.perf_counter() - t_e2e
        if prod is not None:
            tm.producer_busy, tm.producer_blocked, tm.producer_copy = prod.t_busy, prod.t_blocked + prod.t_put, prod.t_copy
        if prod is not None and os.environ.get("QP_PIPELINE_DEBUG"):
            import sys
            print(f"[pipeline] producer waits: ring semaphore {prod.t_sem:.3f} s, previous H2D of the pinned slot {prod.t_sync:.3f} s, full queue "
                  f"{prod.t_put:.3f} s; consumer in get() {tm.consumer_get_wait:.3f} s", file=sys.stderr, flush=True)
        if self.use_gpu and trace and lead:
            self._device_breakdown(tm, origin, trace)
            if self.measure_vit_alone and last_frames is not None and not par.on:
                tm.vit_uncontended = self._vit_alone(last_frames) * G
        tm.layout = self.last_layout
        self.last_timings = tm
        return out

    @staticmethod
    def _device_breakdown(tm: Timings, origin, trace):
        """Event timestamps (ms since `origin`) -> who the main stream waited for between two groups.  The stall in front of group g,
        [end of prefill(g-1), start of prefill(g)], is split at the moment group g's frames finished uploading: before it the GPU
        could not have started ViT(g) (frame wait, the producer's fault); after it the ViT simply was not finished (the tower's)."""
        # an event of this process, or (native ring) the copy's finish time in ms after `origin` as the library measured it
        at = lambda e: e * 1e-3 if isinstance(e, float) else origin.elapsed_time(e) * 1e-3   # noqa: E731  (NaN = the library has no stamp)
        prev_end = 0.0
        tm.group_gaps = []
        for h2d, v0, v1, p0, p1 in trace:
            t_h2d, t_v0, t_v1, t_p0, t_p1 = at(h2d), at(v0), at(v1), at(p0), at(p1)
            stall = max(0.0, t_p0 - prev_end)
            tm.group_gaps.append(stall)
            if t_h2d != t_h2d:                            # upload time unknown: do not blame the tower for what may be a frame wait
                tm.gpu_stall_unknown += stall
            else:
                frames_part = min(stall, max(0.0, t_h2d - prev_end))
                tm.gpu_stall_frames += frames_part
                tm.gpu_stall_vit += stall - frames_part
            tm.gpu_prefill_busy += t_p1 - t_p0
            tm.vit_span += t_v1 - t_v0
            prev_end = t_p1

    def _vit_alone(self, frames) -> float:
        """Seconds for patchify + ViT of one frame group with the GPU otherwise idle (the run itself shares the CUs with the prefill)."""
        torch.cuda.synchronize(self.model.device)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.vit_stream):
            for i in range(3):
                if i == 1:
                    s.record(self.vit_stream)
                rows, grid = self.tower.patchify(frames)
                self.tower.forward(rows, grid)
            e.record(self.vit_stream)
        e.synchronize()
        return s.elapsed_time(e) * 1e-3 / 2
