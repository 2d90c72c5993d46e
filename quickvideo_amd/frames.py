"""Frame sources with the reader contract the reference's overlap pipeline consumes (deepcodec
InterleavedVideoReader [3P] as used in qwen25_lvu_interleaved.py:385-410, 438-442, 513-515):

    vr = Reader(path, num_threads=.., num_intervals=..); len(vr); vr.get_fps()
    vr.height, vr.width, vr.interpolation = H, W, "LANCZOS"     # settable
    vr.process(idx)            # start producing the frames with these indices
    vr.frame_iter = g          # frames per __next__
    next(vr) -> uint8 [g, 3, H, W]   (StopIteration at the end)

There is no FFmpeg / codec in this image, so real containers cannot be decoded; the sources here are
  * synthetic://?frames=F&h=H&w=W&fps=R&seed=S[&pattern=noise|gradient]   seeded frames generated on the fly
    [&decode_h=1080&decode_w=1920[&decode_s=21.3]]   a source that COSTS something, like a decoder: every frame is produced at the
    decode size on one of the reader's `num_threads` workers (QUICKCODEC_CORES) and resized to the requested height x width with
    PIL's LANCZOS filter — the interpolation the reference asks its decoder for (interleaved:438-442).  decode_s = wall seconds
    the whole sampled frame set should cost at this thread count (the reference's figure: 21.3 s of QuickCodec time for a 60-minute
    video on 16 threads, assets/imgs/video_processing_times.png): each worker pads its frame to decode_s * threads / n_frames with a
    sleep when the real work was cheaper (a codec's latency without burning the host); never shortens real work
  * *.npy / *.pt files holding uint8 [F, 3, H, W] (pre-decoded video)
  * a DIRECTORY of image files (frame_000001.jpg ...; sorted by name; optional `fps.txt` holding the frame rate): every requested
    frame is DECODED (PIL: JPEG / PNG / ...) and LANCZOS-resized on the reader's worker threads — a real decode workload without
    FFmpeg (e.g. frames exported once with any tool; the reference keeps such a JPEG frame cache itself, lvu_cache.py:28-49)
  * ONE multi-frame image file (animated GIF / WebP / APNG, multi-page TIFF): the containers PIL decodes by itself — a real file a user can
    point `video_path` at without FFmpeg (AnimatedImageVideoReader)
A real decoder plugs in by implementing the same five members.  Environment knobs QUICKCODEC_CORES /
QUICKCODEC_INTERVALS are read like the reference does (interleaved:391-392) and passed to the reader."""
from __future__ import annotations

import os
import urllib.parse
from typing import Optional

import numpy as np
import torch


def _out_array(out, n, H, W):
    """The array a reader fills: a fresh one, or the caller's (a pinned ring slot) after a shape check."""
    if out is None:
        return np.empty((n, 3, H, W), dtype=np.uint8)
    if tuple(out.shape) != (n, 3, H, W) or out.dtype != np.uint8:
        raise ValueError(f"frames are uint8 {(n, 3, H, W)}, the destination is {out.dtype} {tuple(out.shape)}")
    return out


class VideoReaderBase:
    height: Optional[int] = None
    width: Optional[int] = None
    interpolation: str = "LANCZOS"
    frame_iter: int = 16

    def __init__(self, path: str, num_threads: int = 8, num_intervals: int = 64):
        self.path, self.num_threads, self.num_intervals = path, num_threads, num_intervals
        self._idx = None
        self._cursor = 0

    def __len__(self) -> int: raise NotImplementedError
    def get_fps(self) -> float: raise NotImplementedError
    def _frames(self, idx: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        raise NotImplementedError                                                 # uint8 [len(idx), 3, H, W]; written into `out` when given

    def process(self, idx):
        self._idx = np.asarray(idx, dtype=np.int64)
        self._cursor = 0

    def __iter__(self):
        return self

    def __next__(self) -> torch.Tensor:
        if self._idx is None:
            raise RuntimeError("call process(indices) before iterating")
        if self._cursor >= len(self._idx):
            raise StopIteration
        sel = self._idx[self._cursor:self._cursor + self.frame_iter]
        self._cursor += len(sel)
        return torch.from_numpy(self._frames(sel))

    def pending_indices(self) -> Optional[np.ndarray]:
        """Frame indices of the current selection that have not been handed out yet (None before process()) — what a consumer that
        fetches the frames itself (the native ring's file source) needs, together with `advance`."""
        return None if self._idx is None else self._idx[self._cursor:]

    def advance(self, n_frames: int):
        """Mark the next n_frames of the selection as consumed (they were delivered by other means than next())."""
        if self._idx is not None:
            self._cursor = min(len(self._idx), self._cursor + int(n_frames))

    def next_into(self, dst: np.ndarray) -> int:
        """next() that decodes STRAIGHT into `dst` (uint8 [>= frame_iter, 3, H, W], e.g. a pinned ring slot: no intermediate array, no
        memcpy); returns the number of frames written, 0 at the end of the selection.  Used by the native frame ring (ring.py)."""
        if self._idx is None:
            raise RuntimeError("call process(indices) before iterating")
        sel = self._idx[self._cursor:self._cursor + self.frame_iter]
        if len(sel) == 0:
            return 0
        if len(sel) > dst.shape[0]:
            raise ValueError(f"a group of {len(sel)} frames does not fit a slot of {dst.shape[0]}")
        self._cursor += len(sel)
        out = dst[:len(sel)]
        got = self._frames(sel, out=out)
        if got is not out:                       # a subclass that ignores `out`
            if tuple(got.shape) != tuple(out.shape):
                raise ValueError(f"reader produced frames of shape {tuple(got.shape)}, the slot holds {tuple(out.shape)}")
            np.copyto(out, got)
        return len(sel)


class SyntheticVideoReader(VideoReaderBase):
    """Seeded synthetic video: frame i depends only on (seed, i, H, W, pattern) — any access order gives the same pixels."""

    def __init__(self, path: str, num_threads: int = 8, num_intervals: int = 64):
        super().__init__(path, num_threads, num_intervals)
        q = dict(urllib.parse.parse_qsl(urllib.parse.urlparse(path).query))
        self.total = int(q.get("frames", 64))
        self.src_h, self.src_w = int(q.get("h", 1080)), int(q.get("w", 1920))
        self.fps, self.seed, self.pattern = float(q.get("fps", 2.0)), int(q.get("seed", 1)), q.get("pattern", "noise")
        self.decode_h, self.decode_w = int(q.get("decode_h", 0)), int(q.get("decode_w", 0))
        self.decode_s = float(q.get("decode_s", 0.0))
        self._frame_budget = 0.0          # seconds one worker spends per frame (set in process(): decode_s * threads / n_frames)
        self.work_seconds = 0.0           # thread-seconds of real work (generate + resize), for the report
        self._texture = None
        self._pool = None

    def process(self, idx):
        super().process(idx)
        self.work_seconds = 0.0
        self._frame_budget = self.decode_s * max(self.num_threads, 1) / max(len(self._idx), 1) if self.decode_s > 0 else 0.0

    def __len__(self): return self.total
    def get_fps(self): return self.fps

    def _decoded_frame(self, out, j, i, H, W):
        """Frame i at the decode size -> LANCZOS resize to H x W (PIL releases the GIL; the workers run in parallel)."""
        import time
        from PIL import Image
        t0 = time.perf_counter()
        # "decoded" picture = one seeded noise texture, rolled by the frame index and XORed with a per-frame byte: a pure function of
        # (seed, i) made of large-array numpy ops, which release the GIL (RandomState.randint per frame does not: 8 threads ran
        # no faster than one)
        if self._texture is None:
            self._texture = np.random.RandomState(self.seed % (2 ** 31)).randint(0, 256, (self.decode_h, self.decode_w, 3), dtype=np.uint8)
        src = np.bitwise_xor(np.roll(self._texture, int(i) % self.decode_h, axis=0), np.uint8((int(i) * 37) & 255))
        img = Image.fromarray(src).resize((W, H), Image.LANCZOS)
        out[j] = np.asarray(img).transpose(2, 0, 1)
        dt = time.perf_counter() - t0
        self.work_seconds += dt            # (approximate under threads: float += is not atomic; a report figure only)
        if dt < self._frame_budget:
            time.sleep(self._frame_budget - dt)

    def _frame(self, out, j, i, H, W):
        if self.decode_h and self.decode_w:
            return self._decoded_frame(out, j, i, H, W)
        if self.pattern == "gradient":
            yy = (np.arange(H, dtype=np.uint32)[:, None] * 255 // max(H - 1, 1))
            xx = (np.arange(W, dtype=np.uint32)[None, :] * 255 // max(W - 1, 1))
            base = (yy + xx + 7 * int(i)) % 256
            out[j] = np.stack([base, (base * 3 + 40) % 256, (255 - base)]).astype(np.uint8)
        else:
            out[j] = np.random.RandomState((self.seed * 1_000_003 + int(i)) % (2 ** 31)).randint(0, 256, (3, H, W), dtype=np.uint8)

    def _frames(self, idx, out=None):
        H, W = self.height or self.src_h, self.width or self.src_w
        out = _out_array(out, len(idx), H, W)
        if self.num_threads > 1 and len(idx) > 1 and H * W >= 1 << 16:
            # like the reference's decoder pool (QUICKCODEC_CORES, qwen25_lvu_interleaved.py:385-396): frames of a group in parallel
            # (numpy's generators release the GIL while they fill the buffer)
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.num_threads)
            list(self._pool.map(lambda ji: self._frame(out, ji[0], ji[1], H, W), enumerate(idx)))
        else:
            for j, i in enumerate(idx):
                self._frame(out, j, i, H, W)
        return out


class ArrayVideoReader(VideoReaderBase):
    """Pre-decoded video in a .npy / .pt file: uint8 [F, 3, H, W].  Frames are served at their stored size
    (vr.height/width must match: resizing belongs to a decoder, which this image does not have)."""

    def __init__(self, path: str, num_threads: int = 8, num_intervals: int = 64):
        super().__init__(path, num_threads, num_intervals)
        arr = torch.load(path).numpy() if path.endswith(".pt") else np.load(path, mmap_mode="r")
        assert arr.ndim == 4 and arr.shape[1] == 3 and arr.dtype == np.uint8, "expected uint8 [F, 3, H, W]"
        self.arr = arr
        self.fps = 2.0

    def __len__(self): return self.arr.shape[0]
    def get_fps(self): return self.fps

    def _frames(self, idx, out=None):
        if self.height and self.width and (self.height, self.width) != tuple(self.arr.shape[2:]):
            raise ValueError(f"stored frames are {tuple(self.arr.shape[2:])}, requested {(self.height, self.width)}: no resizer without a codec")
        if out is None:
            return np.ascontiguousarray(self.arr[idx])
        out = _out_array(out, len(idx), *self.arr.shape[2:])
        for j, i in enumerate(idx):
            out[j] = self.arr[int(i)]
        return out

    def raw_layout(self):
        """(path, byte offset of frame 0, bytes per frame) when the frames lie contiguously in a plain file (.npy, C order) — what the
        native ring's built-in file source needs (qp_frame_ring_start_file); None otherwise (.pt)."""
        a = self.arr
        if isinstance(a, np.memmap) and a.flags["C_CONTIGUOUS"] and type(self)._frames is ArrayVideoReader._frames:
            # (a subclass that produces its frames differently is not "the file as it lies")
            if self.height and self.width and (self.height, self.width) != tuple(a.shape[2:]):
                return None                               # _frames() raises for this request: let the caller hear it there
            return str(a.filename), int(a.offset), int(np.prod(a.shape[1:]))
        return None

    def stored_hw(self):
        return tuple(int(v) for v in self.arr.shape[2:])


class ImageFolderVideoReader(VideoReaderBase):
    """A video as a directory of still images (one per frame, sorted by file name).  next() decodes the requested frames with PIL on
    `num_threads` workers (decode and resize release the GIL) and resizes them to vr.height x vr.width with the requested
    interpolation — the work a codec-backed reader does, minus the container."""
    EXT = (".jpg", ".jpeg", ".png", ".bmp", ".webp")

    def __init__(self, path: str, num_threads: int = 8, num_intervals: int = 64):
        super().__init__(path, num_threads, num_intervals)
        self.files = sorted(os.path.join(path, f) for f in os.listdir(path) if f.lower().endswith(self.EXT))
        if not self.files:
            raise ValueError(f"{path!r}: no image files ({', '.join(self.EXT)})")
        fps_file = os.path.join(path, "fps.txt")
        self.fps = float(open(fps_file).read().strip()) if os.path.exists(fps_file) else 2.0
        from PIL import Image
        with Image.open(self.files[0]) as im:
            self.src_w, self.src_h = im.size
        self._pool = None

    def __len__(self): return len(self.files)
    def get_fps(self): return self.fps

    def _one(self, out, j, i, H, W):
        from PIL import Image
        flt = {"LANCZOS": Image.LANCZOS, "BICUBIC": Image.BICUBIC, "BILINEAR": Image.BILINEAR, "NEAREST": Image.NEAREST}.get(
            str(self.interpolation).upper(), Image.LANCZOS)
        with Image.open(self.files[int(i)]) as im:
            im = im.convert("RGB")
            if im.size != (W, H):
                im = im.resize((W, H), flt)
            out[j] = np.asarray(im).transpose(2, 0, 1)

    def _frames(self, idx, out=None):
        H, W = self.height or self.src_h, self.width or self.src_w
        out = _out_array(out, len(idx), H, W)
        if self.num_threads > 1 and len(idx) > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.num_threads)
            list(self._pool.map(lambda ji: self._one(out, ji[0], ji[1], H, W), enumerate(idx)))
        else:
            for j, i in enumerate(idx):
                self._one(out, j, i, H, W)
        return out


class AnimatedImageVideoReader(VideoReaderBase):
    """A video as ONE multi-frame image file — animated GIF / WebP / APNG, multi-page TIFF — decoded with PIL (no FFmpeg in the image, but
    these containers PIL reads itself).  Frame rate from the file's per-frame duration (GIF / WebP / APNG `duration` in ms), else `fps`
    in the query string of the path (`clip.gif?fps=12`), else 10.  Inter-frame coded formats are decoded sequentially up to the highest
    requested index once (process()), then served from memory; every requested frame is resized with the reader's interpolation."""
    EXT = (".gif", ".webp", ".apng", ".png", ".tif", ".tiff")

    def __init__(self, path: str, num_threads: int = 8, num_intervals: int = 64):
        super().__init__(path, num_threads, num_intervals)
        from PIL import Image
        base, _, query = path.partition("?")
        self.file = base
        q = dict(urllib.parse.parse_qsl(query))
        with Image.open(base) as im:
            self.total = int(getattr(im, "n_frames", 1))
            self.src_w, self.src_h = im.size
            dur = im.info.get("duration")
        self.fps = float(q["fps"]) if "fps" in q else (1000.0 / dur if dur else 10.0)
        self._decoded = {}

    def __len__(self): return self.total
    def get_fps(self): return self.fps

    def process(self, idx):
        super().process(idx)
        from PIL import Image
        want = set(int(i) for i in self._idx)
        self._decoded = {}
        with Image.open(self.file) as im:
            for i in range(max(want) + 1 if want else 0):          # sequential: GIF / APNG frames are deltas of their predecessors
                im.seek(i)
                if i in want:
                    self._decoded[i] = im.convert("RGB").copy()

    def _frames(self, idx, out=None):
        from PIL import Image
        H, W = self.height or self.src_h, self.width or self.src_w
        flt = {"LANCZOS": Image.LANCZOS, "BICUBIC": Image.BICUBIC, "BILINEAR": Image.BILINEAR, "NEAREST": Image.NEAREST}.get(
            str(self.interpolation).upper(), Image.LANCZOS)
        out = _out_array(out, len(idx), H, W)
        for j, i in enumerate(idx):
            im = self._decoded[int(i)]
            if im.size != (W, H):
                im = im.resize((W, H), flt)
            out[j] = np.asarray(im).transpose(2, 0, 1)
        return out


def open_video(path, num_threads: Optional[int] = None, num_intervals: Optional[int] = None) -> VideoReaderBase:
    if isinstance(path, VideoReaderBase):
        return path
    # any object with the InterleavedVideoReader contract (deepcodec [3P], qwen25_lvu_interleaved.py:385-410, 438-442, 513-515) is a
    # reader: len(), get_fps(), process(idx), __next__ -> uint8 [g,3,H,W], settable height / width / interpolation / frame_iter
    if not isinstance(path, (str, bytes, os.PathLike)):
        missing = [m for m in ("__len__", "__next__", "get_fps", "process") if not callable(getattr(path, m, None))]
        if missing:
            raise TypeError(f"video reader object lacks {missing}: the contract is len(), get_fps(), process(idx), next() -> uint8 [g,3,H,W] "
                            f"and settable height / width / interpolation / frame_iter")
        return path
    nt = num_threads if num_threads is not None else int(os.environ.get("QUICKCODEC_CORES", "8"))
    ni = num_intervals if num_intervals is not None else int(os.environ.get("QUICKCODEC_INTERVALS", "64"))
    p = str(path)
    if p.startswith("synthetic://"):
        return SyntheticVideoReader(p, nt, ni)
    if p.endswith((".npy", ".pt")):
        return ArrayVideoReader(p, nt, ni)
    if os.path.isdir(p):
        return ImageFolderVideoReader(p, nt, ni)
    if p.partition("?")[0].lower().endswith(AnimatedImageVideoReader.EXT) and os.path.isfile(p.partition("?")[0]):
        return AnimatedImageVideoReader(p, nt, ni)
    raise ValueError(f"cannot open {p!r}: this build has no video codec (no FFmpeg in the image); use synthetic://..., a "
                     f".npy/.pt file of uint8 [F,3,H,W] frames, a directory of frame images, an animated GIF / WebP / APNG / multi-page "
                     f"TIFF, or pass a reader object with the InterleavedVideoReader contract")


def smart_nframes(total_frames: int, video_fps: float, nframes: Optional[int] = None, fps: Optional[float] = None) -> int:
    """Keyword form of planner.smart_nframes (the reference's function takes the video entry of the message, qwen25_lvu.py:402-442)."""
    from .planner import smart_nframes as _entry_form
    if nframes is not None and fps is not None:
        raise ValueError("Only accept either `fps` or `nframes`")
    return int(_entry_form({"nframes": nframes} if nframes is not None else ({"fps": fps} if fps is not None else {}), total_frames, video_fps))
