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


def video_frame_size(n_frames: int, height: int, width: int, max_pixels: Optional[int] = None, min_pixels: Optional[int] = None):
    """Per-frame resize target under the reference's pixel budget (qwen25_lvu.py:292-306)."""
    vmin, vmax, vtot, ff = 128 * 28 * 28, 768 * 28 * 28, 24576 * 28 * 28, 2
    mn = vmin if min_pixels is None else min_pixels
    mx = max(min(vmax, vtot / n_frames * ff), int(mn * 1.05)) if max_pixels is None else max_pixels
    return smart_resize(height, width, factor=28, min_pixels=mn, max_pixels=mx)
