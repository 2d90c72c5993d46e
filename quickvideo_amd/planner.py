"""Host-side integer planning of the group loop: group sizes, M-RoPE position ids, frame-size budget.
Restates lvu/models/qwen25_lvu.py:609-665 (planner) and the [3P] get_rope_index / smart_resize arithmetic."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np


@dataclass
class GroupPlan:
    tokens: List[int]                       # q_len per group; group 0 also owns the text prefix (qwen25_lvu.py:665)
    grid_thw: List[Tuple[int, int, int]]    # per-group ViT grid (qwen25_lvu.py:633-640)
    pixel_rows: List[int]                   # rows of pixel_values_videos per group (:641-642)
    frames: List[int]
    past_len_after: int                     # where the prompt tail starts
    tail_len: int


def plan_groups(n_frames: int, video_group_size: Optional[int], grid_h: int, grid_w: int, prefix_len: int, total_len: int,
                temporal_patch_size: int = 2, merge: int = 2) -> GroupPlan:
    if video_group_size is None:
        raise TypeError("video_group_size must be an int (0 = single group)")      # reference: None % 2 -> TypeError (:625)
    grid_t = n_frames // temporal_patch_size
    n_video_tokens = grid_t * (grid_h // merge) * (grid_w // merge)
    rows_total = grid_t * grid_h * grid_w
    gs = video_group_size
    if gs % temporal_patch_size != 0:
        gs += temporal_patch_size - (gs % temporal_patch_size)
    if gs > 0:
        frames = [min(gs, n_frames - s) for s in range(0, n_frames, gs)]
        if not all(f % 2 == 0 for f in frames):
            raise AssertionError("The video group size should be even.")
        tokens = [int(n_video_tokens * (f / n_frames)) for f in frames]
        grids = [((f - 1) // temporal_patch_size + 1, grid_h, grid_w) for f in frames]
        per = round((gs / n_frames) * rows_total)
        pixel_rows = [min(per, rows_total - s) for s in range(0, rows_total, per)]
    else:
        frames, tokens, grids, pixel_rows = [n_frames], [n_video_tokens], [(grid_t, grid_h, grid_w)], [rows_total]
    tokens[0] += prefix_len
    past = sum(tokens)
    if not past < total_len:
        raise AssertionError("The past length should be less than the final input length.")
    return GroupPlan(tokens, grids, pixel_rows, frames, past, total_len - past)


def temporal_ids(t: int, temporal_scale: float = 1.0, second_per_grid_t: Optional[float] = None,
                 tokens_per_second: Optional[float] = None) -> np.ndarray:
    """Temporal M-RoPE id of each of the t temporal patches.  Qwen2-VL: 0..t-1.  Qwen2.5-VL (get_rope_index [3P]):
    `(arange(t) * second_per_grid_t * tokens_per_second).long()` where second_per_grid_t is a FLOAT32 tensor element
    (`second_per_grid_ts` goes through torch.tensor) — so both products are rounded to fp32 before the truncation.  Evaluating
    the same expression in float64 differs by one whenever a product lands on an integer in one precision and just below it in
    the other (about 2 % of (video length, nframes) pairs; e.g. 24 fps, 28416 frames, 768 sampled)."""
    if second_per_grid_t is not None:
        spg, tps = np.float32(second_per_grid_t), np.float32(tokens_per_second)
        return ((np.arange(t, dtype=np.int64).astype(np.float32) * spg) * tps).astype(np.int64)
    if temporal_scale == 1.0:
        return np.arange(t, dtype=np.int64)
    return (np.arange(t, dtype=np.float64) * temporal_scale).astype(np.int64)


def mrope_positions(prefix_len: int, grid_thw: Tuple[int, int, int], tail_len: int, merge: int = 2,
                    temporal_scale: float = 1.0, second_per_grid_t: Optional[float] = None,
                    tokens_per_second: Optional[float] = None) -> Tuple[np.ndarray, int]:
    """int64 [3, T] ids for <prefix text><video><tail text> and rope_delta (transformers 4.50 get_rope_index rule:
    text after the video resumes at max(position)+1).  Qwen2.5-VL: pass second_per_grid_t (= temporal_patch / sampled fps)
    and tokens_per_second; the temporal stream is then computed with HF's float32 arithmetic (temporal_ids)."""
    t, h, w = grid_thw[0], grid_thw[1] // merge, grid_thw[2] // merge
    pre = np.tile(np.arange(prefix_len, dtype=np.int64), (3, 1))
    ti = temporal_ids(t, temporal_scale, second_per_grid_t, tokens_per_second)
    vid = np.stack([np.repeat(ti, h * w), np.tile(np.repeat(np.arange(h, dtype=np.int64), w), t),
                    np.tile(np.arange(w, dtype=np.int64), t * h)]) + prefix_len
    st = int(vid.max()) + 1 if vid.size else prefix_len
    tail = np.tile(np.arange(tail_len, dtype=np.int64), (3, 1)) + st
    pos = np.concatenate([pre, vid, tail], axis=1)
    return pos, int(pos.max()) + 1 - pos.shape[1]


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280):
    """qwen-vl-utils smart_resize [3P]."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = max(factor, round(height / factor) * factor)
    w_bar = max(factor, round(width / factor) * factor)
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = math.floor(height / beta / factor) * factor
        w_bar = math.floor(width / beta / factor) * factor
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


# qwen-vl-utils 0.0.10 constants [3P] (uv.lock:1046-1047; imported by `from qwen_vl_utils.vision_process import *`, qwen25_lvu.py:26) and
# the reference's own override of FPS_MAX_FRAMES (qwen25_lvu.py:27).  VIDEO_TOTAL_PIXELS is SURVEY §8d's figure; like the [3P]
# package, the environment variable VIDEO_MAX_PIXELS overrides it.
IMAGE_FACTOR, FRAME_FACTOR = 28, 2
VIDEO_MIN_PIXELS, VIDEO_MAX_PIXELS = 128 * 28 * 28, 768 * 28 * 28
FPS, FPS_MIN_FRAMES, FPS_MAX_FRAMES = 2.0, 4, 100_000


def video_total_pixels() -> int:
    import os
    return int(float(os.environ.get("VIDEO_MAX_PIXELS", 24576 * 28 * 28)))


def smart_nframes(ele: dict, total_frames: int, video_fps: float) -> int:
    """Frames sampled from a video of `total_frames` at `video_fps`, from the VIDEO ENTRY of the chat message — the reference's own
    smart_nframes (qwen25_lvu.py:402-442, twin qwen25_lvu_interleaved.py:343-383): `nframes` (rounded to a multiple of 2, clamped to the
    video's length) XOR `fps` (default 2.0) with `min_frames` / `max_frames`.  Pinned by GV4 (tests/golden/gv4_video_plan.json)."""
    assert not ("fps" in ele and "nframes" in ele), "Only accept either `fps` or `nframes`"
    ff = FRAME_FACTOR
    if "nframes" in ele:
        nframes = round(ele["nframes"] / ff) * ff
        nframes = min(nframes, total_frames)
        nframes -= nframes % ff
    else:
        fps = ele.get("fps", FPS)
        min_frames = math.ceil(ele.get("min_frames", FPS_MIN_FRAMES) / ff) * ff
        max_frames = math.floor(ele.get("max_frames", min(FPS_MAX_FRAMES, total_frames)) / ff) * ff
        nframes = total_frames / video_fps * fps
        nframes = min(min(max(nframes, min_frames), max_frames), total_frames)
        nframes = math.floor(nframes / ff) * ff
    if not (ff <= nframes and nframes <= total_frames):
        raise ValueError(f"nframes should in interval [{ff}, {total_frames}], but got {nframes}.")
    return nframes


def video_pixel_budget(nframes: int, ele: Optional[dict] = None):
    """(min_pixels, max_pixels) per frame (qwen25_lvu.py:292-298, twins :351-357 and interleaved:416-422): the budget
    `max(min(VIDEO_MAX_PIXELS, total_pixels / nframes * 2), int(min_pixels * 1.05))` is a LIMIT — a `max_pixels` in the video entry can
    only lower it (the reference logs a warning when it asks for more); `total_pixels` in the entry moves the budget itself."""
    ele = ele or {}
    total_pixels = ele.get("total_pixels", video_total_pixels())
    min_pixels = ele.get("min_pixels", VIDEO_MIN_PIXELS)
    limit = max(min(VIDEO_MAX_PIXELS, total_pixels / nframes * FRAME_FACTOR), int(min_pixels * 1.05))
    wanted = ele.get("max_pixels", limit)
    if wanted > limit:
        import logging
        logging.getLogger(__name__).warning(f"The given max_pixels[{wanted}] exceeds limit[{limit}].")
    return min_pixels, min(wanted, limit)


def video_frame_size(n_frames: int, height: int, width: int, ele: Optional[dict] = None):
    """Per-frame resize target (qwen25_lvu.py:292-306, :358-372; interleaved:416-436): `resized_height` + `resized_width` in the video
    entry win (rounded to multiples of 28 under the IMAGE pixel limits, like the reference's call without min/max arguments);
    otherwise smart_resize of the source size under video_pixel_budget."""
    ele = ele or {}
    min_pixels, max_pixels = video_pixel_budget(n_frames, ele)           # evaluated first, like the reference (its warning fires either way)
    if "resized_height" in ele and "resized_width" in ele:
        return smart_resize(ele["resized_height"], ele["resized_width"], factor=IMAGE_FACTOR, min_pixels=4 * 28 * 28, max_pixels=16384 * 28 * 28)
    return smart_resize(height, width, factor=IMAGE_FACTOR, min_pixels=min_pixels, max_pixels=max_pixels)


VIDEO_ENTRY_KEYS = ("fps", "nframes", "min_frames", "max_frames", "min_pixels", "max_pixels", "total_pixels", "resized_height", "resized_width")


def video_entry_from_config(video, fps=None, num_frames=None, extra_kwargs=None) -> dict:
    """The video entry run_lvu_model builds from the LVUConfig (qwen25_lvu.py:504-536): max_pixels / min_pixels from extra_kwargs,
    `fps` if set, else `nframes`."""
    extra_kwargs = extra_kwargs or {}
    ele = {"type": "video", "video": video}
    for k in ("max_pixels", "min_pixels"):
        if extra_kwargs.get(k) is not None:
            ele[k] = extra_kwargs[k]
    if fps is not None:
        ele["fps"] = fps
    elif num_frames is not None:
        ele["nframes"] = num_frames
    else:
        raise ValueError("Either fps or num_frames should be set.")
    return ele
