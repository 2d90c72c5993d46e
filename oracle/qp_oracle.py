"""CPU oracle for the QuickPrefill hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``quickvideo_amd/`` imports it: the product path is the HIP library
behind ``include/quickprefill.h`` and fails loudly when that library is missing.

It restates, in numpy (integer / byte work) and torch-CPU (floating-point model math), the
algorithm of the reference's group-chunked prefill with key-L2-norm KV pruning.  Every function
cites the reference file:line (paths relative to /root/reference) it follows.

Pinning status (see oracle/make_golden.py, tests/golden/*.npz|json, tests/test_oracle_golden.py):
  * effective_k, key-norm select, KV compaction, hidden-state pruning hand-off: pinned against the
    reference's own ``lvu/utils.py`` functions imported in the build container (golden GV1-GV3).
  * group planner: restated from lvu/models/qwen25_lvu.py:623-665 (not separately importable);
    pinned by hand-derived cases + the composite end-to-end run (GV4/GV5).
  * decoder math (RMSNorm, QKV, M-RoPE, bottom-right-causal attention, MLP): the reference gets it
    from transformers==4.50.0 + flash-attn (absent from /root/reference; uv.lock:1380-1381).  It is
    restated from the published model definition and pinned against the *installed* transformers
    5.15 Qwen2-VL modules driven with the reference's post_process_kv_cache (composite oracle,
    golden GV5).  The reference itself holds no test for any of this: "parity unpinned by the
    reference's own tests" — our pins are outputs of the reference's functions run here.

Canonical definitions where the reference is under-specified (SURVEY.md §7 hard parts):
  * key norm  = bf16_rne( sqrt_f32( S ) ),  S = fp32 sum of squares in the FIXED order of
    ``key_sumsq_heads`` + ``key_norms_bf16`` below (8-element sequential chunks, xor-butterfly over the
    16 chunks of a head row, heads added in ascending order).  torch's own CPU/CUDA reductions use
    other orders; on random data they agree with this one except on ~1e-5 of rows (measured and
    recorded in the golden file).
  * tie rule  = k smallest by (bf16 norm, then lowest index) — i.e. a STABLE ascending sort.  That is
    what the reference does on its real deployment (torch CUDA sort is a stable radix sort); torch's
    CPU ``argsort(stable=False)`` (introsort) orders ties differently, so CPU-reference fixtures with a
    tie straddling the k-th place are compared through the threshold property instead.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# bf16 helpers (numpy has no bf16: carry it as uint16 bit patterns)
# --------------------------------------------------------------------------------------

def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (uint16); NaN stays NaN (quiet)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = (u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)
    nan = np.isnan(x)
    out = rounded.astype(np.uint16)
    out[nan] = ((u[nan] >> np.uint32(16)) | np.uint32(0x0040)).astype(np.uint16)
    return out


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def torch_bf16_to_bits(t: torch.Tensor) -> np.ndarray:
    assert t.dtype == torch.bfloat16
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def bits_to_torch_bf16(b: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(b, dtype=np.uint16).view(np.int16)).view(torch.bfloat16)


# --------------------------------------------------------------------------------------
# a8 / a6: effective-k rule                                   lvu/utils.py:231-255
# --------------------------------------------------------------------------------------

def effective_k(q_len: int, top_k: Optional[int], top_p: Optional[float], decay_type: Optional[str],
                decay_factor: Optional[float], layer_idx: int, total_layers: int, enable: bool = True,
                top_k_starting_layer: Optional[int] = None) -> Optional[int]:
    """Number of new tokens kept by this layer, or None when the layer does not prune.

    utils.py:241-242 top_p clamp (IEEE-double multiply then truncation), :244-251 decay,
    :252-255 early-outs.  ``top_k_starting_layer`` follows the *intended* rule (skip layers below
    it); the reference raises AttributeError there (utils.py:253 reads a field that does not exist).
    """
    if top_p is not None and top_p >= 0:
        top_k = min((top_k or q_len), int(q_len * top_p))
    if not decay_type:
        pass
    elif decay_type == "linear":
        top_k = top_k - int(top_k * (layer_idx / total_layers))
    elif decay_type == "exponential":
        top_k = int(top_k * (decay_factor ** layer_idx))
    else:
        raise ValueError(f"Unknown top_k_decay_type: {decay_type}")
    if not enable or not top_k or top_k <= 0 or q_len <= top_k:
        return None
    if isinstance(top_k_starting_layer, int) and top_k_starting_layer > 0 and layer_idx < top_k_starting_layer:
        return None
    return int(top_k)


# --------------------------------------------------------------------------------------
# a7: key-norm scoring + k-smallest select                     lvu/utils.py:133-136, 190-194
# --------------------------------------------------------------------------------------

def key_sumsq_heads(k_bits: np.ndarray) -> np.ndarray:
    """Per-head fp32 sum of squares in the canonical order.

    k_bits: uint16 [Hkv, n, D] (bf16 patterns), D in {64, 128}.  Returns float32 [Hkv, n].
    Order: each run of 8 consecutive elements is accumulated left to right (x*x is exact in fp32,
    so fma and mul+add agree); the D/8 chunk partials are then combined by an xor-butterfly with
    strides 1,2,4,8 — exactly what 16 lanes of a wavefront do with row-wise cross-lane adds.
    """
    hkv, n, d = k_bits.shape
    assert d % 8 == 0 and (d // 8) in (8, 16), "head_dim must be 64 or 128"
    x = bf16_bits_to_f32(k_bits).reshape(hkv, n, d // 8, 8)
    sq = x * x  # exact
    part = sq[..., 0].copy()
    for e in range(1, 8):
        part = part + sq[..., e]  # float32 adds, sequential
    c = d // 8
    idx = np.arange(c)
    stride = 1
    while stride < c:
        part = part + part[..., idx ^ stride]
        stride *= 2
    return np.ascontiguousarray(part[..., 0], dtype=np.float32)


def key_norms_bf16(head_sumsq: np.ndarray) -> np.ndarray:
    """float32 [Hkv, n] per-head sums -> uint16 [n] bf16 norm of the concatenated heads.

    Heads are added in ascending order ((h0+h1)+h2)+...; sqrt is IEEE correctly rounded fp32;
    result rounded to nearest-even bf16 (utils.py:134-135: the norm of a bf16 tensor is bf16).
    """
    s = head_sumsq[0].astype(np.float32).copy()
    for h in range(1, head_sumsq.shape[0]):
        s = s + head_sumsq[h]
    return f32_to_bf16_bits(np.sqrt(s, dtype=np.float32))


def select_k_smallest(norm_bits: np.ndarray, k: int) -> np.ndarray:
    """Ascending int32 indices of the k smallest norms, ties -> lowest index first.

    utils.py:136 ``argsort(descending=False)[:k]`` then :191-194/:284 mask -> nonzero (ascending
    position order).  Non-negative bf16 patterns order like unsigned ints; NaN sorts last, as in torch.
    """
    keys = norm_bits.astype(np.int64) if norm_bits.dtype == np.uint16 else norm_bits
    order = np.argsort(keys, kind="stable")[:k]
    return np.sort(order).astype(np.int32)


def select_k_largest(norm_bits: np.ndarray, k: int) -> np.ndarray:
    """Ascending int32 indices of the k LARGEST norms, ties -> lowest index first (the reference's ``key_norms`` /
    ``vector_norms``: utils.py:117-131 ``argsort(descending=True)[:k]`` in its stable form, then mask -> nonzero)."""
    keys = (0xFFFF - norm_bits.astype(np.int64)) if norm_bits.dtype == np.uint16 else -norm_bits
    order = np.argsort(keys, kind="stable")[:k]
    return np.sort(order).astype(np.int32)


# norm-based predict types (utils.py:117-136): name -> (norm source: 0 key rows / 1 value rows, order: 0 smallest / 1 largest)
NORM_PRUNE_MODES = {"key_norms_small": (0, 0), "key_norms": (0, 1), "vector_norms_small": (1, 0), "vector_norms": (1, 1)}


def select_threshold(norm_bits: np.ndarray, k: int) -> Tuple[int, int, int]:
    """(tau, n_less, n_equal): k-th smallest pattern, #patterns < tau, #patterns == tau."""
    srt = np.sort(norm_bits.astype(np.int64))
    tau = int(srt[k - 1])
    return tau, int((norm_bits < tau).sum()), int((norm_bits == tau).sum())


# --------------------------------------------------------------------------------------
# a6: KV pruning of the group tail                              lvu/utils.py:257-342
# --------------------------------------------------------------------------------------

def prune_tail(k_cache: np.ndarray, v_cache: np.ndarray, past_len: int, n: int, k: int):
    """In-place restatement of post_process_kv_cache's KV part on a pre-allocated arena.

    k_cache/v_cache: uint16 [Hkv, capacity, D].  Rows [past_len, past_len+n) are the group's new
    tokens (utils.py:266-271).  After the call rows [past_len, past_len+k) hold the kept tokens in
    original order (utils.py:284-288, 333-336).  Returns (kept_idx int32[k], norm_bits uint16[n]).
    """
    new_k = k_cache[:, past_len:past_len + n]
    norms = key_norms_bf16(key_sumsq_heads(new_k))
    idx = select_k_smallest(norms, k)
    k_cache[:, past_len:past_len + k] = new_k[:, idx]
    v_cache[:, past_len:past_len + k] = v_cache[:, past_len:past_len + n][:, idx]
    return idx, norms


# --------------------------------------------------------------------------------------
# a1: group planner                                             lvu/models/qwen25_lvu.py:609-665
# --------------------------------------------------------------------------------------

@dataclass
class GroupPlan:
    tokens: List[int]              # q_len of each group (group 0 includes the text prefix, :665)
    grid_thw: List[Tuple[int, int, int]]
    pixel_rows: List[int]          # rows of pixel_values_videos fed to the ViT per group
    frames: List[int]              # frames per group
    past_len_after: int            # sequence position where the prompt tail starts
    tail_len: int


def plan_groups(n_frames: int, video_group_size: int, grid_h: int, grid_w: int, prefix_len: int,
                total_len: int, temporal_patch_size: int = 2, merge: int = 2) -> GroupPlan:
    """qwen25_lvu.py:623-647 + :665.  grid_h/grid_w are ViT patch grid sizes (pre-merge)."""
    grid_t = n_frames // temporal_patch_size
    n_video_tokens = grid_t * (grid_h // merge) * (grid_w // merge)
    pixel_rows_total = grid_t * grid_h * grid_w
    gs = video_group_size
    if gs is not None and gs % temporal_patch_size != 0:       # :625-626
        gs += temporal_patch_size - (gs % temporal_patch_size)
    if gs is not None and gs > 0:
        frames = [min(gs, n_frames - s) for s in range(0, n_frames, gs)]   # tensor.split(gs), :628
        assert all(f % 2 == 0 for f in frames), "The video group size should be even."   # :629
        tokens = [int(n_video_tokens * (f / n_frames)) for f in frames]    # :630
        grids = [((f - 1) // temporal_patch_size + 1, grid_h, grid_w) for f in frames]  # :633-640
        rows_per = round((gs / n_frames) * pixel_rows_total)               # :641
        pixel_rows = [min(rows_per, pixel_rows_total - s) for s in range(0, pixel_rows_total, rows_per)]  # :642
    else:                                                      # :643-647 (video_group_size == 0)
        frames, tokens, grids, pixel_rows = [n_frames], [n_video_tokens], [(grid_t, grid_h, grid_w)], [pixel_rows_total]
    tokens = list(tokens)
    tokens[0] += prefix_len                                    # :665
    past = sum(tokens)
    assert past < total_len, "The past length should be less than the final input length."   # :718
    return GroupPlan(tokens, grids, pixel_rows, frames, past, total_len - past)


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    """qwen-vl-utils 0.0.10 smart_resize (uv.lock:1046-1047; call site qwen25_lvu.py:292-306)."""
    def round_by(x): return round(x / factor) * factor
    def ceil_by(x): return math.ceil(x / factor) * factor
    def floor_by(x): return math.floor(x / factor) * factor
    h_bar = max(factor, round_by(height)); w_bar = max(factor, round_by(width))
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = floor_by(height / beta); w_bar = floor_by(width / beta)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = ceil_by(height * beta); w_bar = ceil_by(width * beta)
    return h_bar, w_bar


def smart_nframes(ele: dict, total_frames: int, video_fps: float, fps_max_frames: int = 100_000) -> int:
    """qwen25_lvu.py:402-442 (twin qwen25_lvu_interleaved.py:343-383) with FPS_MAX_FRAMES = 100_000 (qwen25_lvu.py:27) and the
    qwen-vl-utils constants FRAME_FACTOR=2, FPS=2.0, FPS_MIN_FRAMES=4.  Pinned by GV4."""
    assert not ("fps" in ele and "nframes" in ele), "Only accept either `fps` or `nframes`"
    if "nframes" in ele:
        n = round(ele["nframes"] / 2) * 2                      # round_by_factor
        n = min(n, total_frames)
        n -= n % 2
    else:
        fps = ele.get("fps", 2.0)
        lo = math.ceil(ele.get("min_frames", 4) / 2) * 2       # ceil_by_factor
        hi = math.floor(ele.get("max_frames", min(fps_max_frames, total_frames)) / 2) * 2
        n = total_frames / video_fps * fps
        n = min(min(max(n, lo), hi), total_frames)
        n = math.floor(n / 2) * 2
    if not (2 <= n <= total_frames):
        raise ValueError(f"nframes should in interval [2, {total_frames}], but got {n}.")
    return n


def video_frame_size(n_frames: int, height: int, width: int, ele: Optional[dict] = None, total_pixels_default: int = 24576 * 28 * 28):
    """Pixel budget + resize target of fetch_video (qwen25_lvu.py:292-306, :351-372; interleaved:416-436): VIDEO_MIN_PIXELS=128*28*28,
    VIDEO_MAX_PIXELS=768*28*28, VIDEO_TOTAL_PIXELS=24576*28*28, FRAME_FACTOR=2; `max_pixels` of the entry is CLAMPED to the budget
    (:296-298), `total_pixels` / `min_pixels` of the entry replace the defaults, `resized_height/width` bypass the budget.  Pinned by GV4."""
    ele = ele or {}
    vmax, ff = 768 * 28 * 28, 2
    total = ele.get("total_pixels", total_pixels_default)
    mn = ele.get("min_pixels", 128 * 28 * 28)
    mx = max(min(vmax, total / n_frames * ff), int(mn * 1.05))
    mx = min(ele.get("max_pixels", mx), mx)
    if "resized_height" in ele and "resized_width" in ele:
        return smart_resize(ele["resized_height"], ele["resized_width"], factor=28, min_pixels=4 * 28 * 28, max_pixels=16384 * 28 * 28)
    return smart_resize(height, width, factor=28, min_pixels=mn, max_pixels=mx)


# --------------------------------------------------------------------------------------
# a12 [3P]: M-RoPE index (transformers 4.50.0 Qwen2-VL / Qwen2.5-VL get_rope_index),
# call site qwen25_lvu.py:613-619.  One video, batch 1, no padding.
# --------------------------------------------------------------------------------------

def mrope_positions(prefix_len: int, grid_thw: Tuple[int, int, int], tail_len: int, merge: int = 2,
                    temporal_scale: float = 1.0) -> Tuple[np.ndarray, int]:
    """int64 [3, T] position ids for  <prefix text> <video tokens> <tail text>  and rope_delta.

    temporal_scale = second_per_grid_t * tokens_per_second for Qwen2.5-VL (integer-truncated, as
    ``.long()`` does there); 1.0 for Qwen2-VL.  Text after the video starts at max(position)+1.
    """
    t, h, w = grid_thw[0], grid_thw[1] // merge, grid_thw[2] // merge
    pre = np.tile(np.arange(prefix_len, dtype=np.int64), (3, 1))
    ti = (np.arange(t, dtype=np.float64) * temporal_scale).astype(np.int64) if temporal_scale != 1.0 else np.arange(t, dtype=np.int64)
    tt = np.repeat(ti, h * w)
    hh = np.tile(np.repeat(np.arange(h, dtype=np.int64), w), t)
    ww = np.tile(np.arange(w, dtype=np.int64), t * h)
    vid = np.stack([tt, hh, ww]) + prefix_len
    st = int(vid.max()) + 1 if vid.size else prefix_len
    tail = np.tile(np.arange(tail_len, dtype=np.int64), (3, 1)) + st
    pos = np.concatenate([pre, vid, tail], axis=1)
    delta = int(pos.max()) + 1 - pos.shape[1]
    return pos, delta


# --------------------------------------------------------------------------------------
# a3 / a4 / a12: decoder math on torch-CPU
# --------------------------------------------------------------------------------------

@dataclass
class TextSpec:
    hidden: int
    n_heads: int
    n_kv_heads: int
    head_dim: int
    intermediate: int
    n_layers: int
    vocab: int
    rope_theta: float = 1_000_000.0
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    rms_eps: float = 1e-6
    tie_embeddings: bool = False


@dataclass
class PruneCfg:
    """The LVUConfig fields the hot path reads (lvu/lvu_config.py:3-33)."""
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    top_k_decay_type: Optional[str] = None
    top_k_decay_factor: Optional[float] = None
    enable: bool = True
    prefill_prune_starting_layer: Optional[int] = None
    top_k_starting_layer: Optional[int] = None
    top_k_predict_type: str = "key_norms_small"       # or key_norms / vector_norms_small / vector_norms (utils.py:117-136),
                                                      # or query_attention_weights[_by_value_norm] (utils.py:55-62; query_based)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2RMSNorm [3P]: fp32 variance, cast back, then weight multiply (qwen25_lvu.py:169,196)."""
    dt = x.dtype
    xf = x.float()
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(dt)


def mrope_cos_sin(pos: torch.Tensor, spec: TextSpec, dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """pos int64 [3, n] -> cos, sin [n, head_dim] in ``dtype``, M-RoPE sections already merged.

    Qwen2VLRotaryEmbedding.forward [3P] (fp32 outer product, cat(freqs,freqs), cos/sin, cast) then the
    section select of apply_multimodal_rotary_pos_emb [3P] (qwen25_lvu.py:51-54).
    """
    d = spec.head_dim
    inv_freq = 1.0 / (spec.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    freqs = pos.float()[:, :, None] * inv_freq[None, None, :]            # [3, n, d/2]
    emb = torch.cat((freqs, freqs), dim=-1)                              # [3, n, d]
    cos, sin = emb.cos().to(dtype), emb.sin().to(dtype)
    sec = list(spec.mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [H, n, D]; cos/sin [n, D].  (x*cos) + (rotate_half(x)*sin), each op rounded in x.dtype."""
    return (x * cos[None]) + (rotate_half(x) * sin[None])


def attention_bottom_right(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """q [Hq, n, D]; k, v [Hkv, P+n, D] -> [n, Hq, D].  Query i sees keys j <= P+i — flash-attn's
    bottom-right-aligned causal mask (qwen25_lvu.py:102-112), GQA by head grouping (repeat_kv, :61-62).
    Scores/softmax in fp32, probabilities cast to the input dtype before P@V like HF's eager path."""
    hq, n, d = q.shape
    hkv, kv, _ = k.shape
    g = hq // hkv
    p = kv - n
    mask = torch.arange(kv)[None, :] <= (torch.arange(n)[:, None] + p)
    out = torch.empty(n, hq, d, dtype=q.dtype)
    for h in range(hq):
        s = (q[h].float() @ k[h // g].float().T) * scale
        s = s.masked_fill(~mask, float("-inf"))
        pr = torch.softmax(s, dim=-1)
        pr = torch.nan_to_num(pr, nan=0.0)             # a query that sees no key at all (kv < n: query-based groups) -> 0, like flash-attn
        out[:, h] = (pr @ v[h // g].float()).to(q.dtype)
    return out


def query_attention_scores(q_prompt: torch.Tensor, k_group: torch.Tensor) -> torch.Tensor:
    """The query-based scoring of LVUCache.update (lvu_cache.py:100-116), op for op: q_prompt [Hq, m, D] = the RoPE'd queries of
    the prompt tokens appended to the group, k_group [Hkv, n, D] = the group's own RoPE'd keys (NOT the past).  Scores q.k^T/sqrt(D)
    in the model dtype, softmax over the n keys in fp32, cast back, summed over the m queries, averaged over the heads -> [n]."""
    hq, m, d = q_prompt.shape
    kr = k_group.repeat_interleave(hq // k_group.shape[0], dim=0)          # repeat_kv
    a = torch.einsum("hqd,hkd->hqk", q_prompt, kr) / (d ** 0.5)
    p = torch.softmax(a, dim=-1, dtype=torch.float32).to(q_prompt.dtype)
    return p.sum(-2).mean(0)


def query_score_keys(score: torch.Tensor, v_group: Optional[torch.Tensor] = None) -> np.ndarray:
    """Sort keys (uint16 patterns, larger = kept first) for predict types query_attention_weights (utils.py:55-57) and, with
    v_group [Hkv, n, D], query_attention_weights_by_value_norm (utils.py:58-62: score * ||v_t|| over all kv heads, bf16 product).
    Scores are non-negative, so the bf16 bit pattern orders like the value; the reference's argsort(descending=True) on its
    deployment device is stable, i.e. ties -> lowest index = select_k_largest."""
    if v_group is not None:
        vn = v_group.transpose(0, 1).flatten(1, 2).norm(2, dim=-1)
        score = score * vn
    if score.dtype == torch.bfloat16:
        return torch_bf16_to_bits(score)
    return score.float().numpy()


class OracleCache:
    """Per-layer K/V with append (lvu_cache.py:90-98 default path: HF DynamicCache.update = cat)."""

    def __init__(self, n_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers   # [Hkv, len, D]
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers

    def append(self, layer: int, k: torch.Tensor, v: torch.Tensor):
        self.k[layer] = k if self.k[layer] is None else torch.cat([self.k[layer], k], dim=1)
        self.v[layer] = v if self.v[layer] is None else torch.cat([self.v[layer], v], dim=1)
        return self.k[layer], self.v[layer]

    def length(self, layer: int) -> int:
        return 0 if self.k[layer] is None else self.k[layer].shape[1]


def _norm_keys_of(k_new: torch.Tensor) -> np.ndarray:
    """Sort keys for the select.  bf16 model (the product dtype): canonical bf16 norm patterns.
    fp32 model (tolerance tests only): fp32 norms, as the reference computes the norm in the key
    dtype (utils.py:134-135) — ties are then practically absent."""
    if k_new.dtype == torch.bfloat16:
        return key_norms_bf16(key_sumsq_heads(torch_bf16_to_bits(k_new)))
    x = k_new.transpose(0, 1).flatten(1, 2).double().numpy()
    return np.sqrt((x * x).sum(-1)).astype(np.float32)


def decoder_layer(h: torch.Tensor, w: dict, layer: int, spec: TextSpec, cache: OracleCache,
                  cos: torch.Tensor, sin: torch.Tensor, k_keep: Optional[int],
                  prune_hidden: bool = False, trace: Optional[dict] = None, predict_type: str = "key_norms_small", prompt_len: int = 0):
    """One patched decoder layer (qwen25_lvu.py:122-212) on h [n, d].

    Returns (h_out, kept_idx or None).  When ``prune_hidden`` (prune_for_next_layer,
    lvu_config.py:50-55) the hidden rows / cos / sin handed to the MLP and to later layers are the
    kept ones (utils.py:292-331, 344-372; qwen25_lvu.py:163-165, 200-202).
    """
    n = h.shape[0]
    p = f"layers.{layer}."
    x = rmsnorm(h, w[p + "input_layernorm.weight"], spec.rms_eps)
    q = torch.nn.functional.linear(x, w[p + "q_proj.weight"], w[p + "q_proj.bias"])
    k = torch.nn.functional.linear(x, w[p + "k_proj.weight"], w[p + "k_proj.bias"])
    v = torch.nn.functional.linear(x, w[p + "v_proj.weight"], w[p + "v_proj.bias"])
    q = q.view(n, spec.n_heads, spec.head_dim).transpose(0, 1)
    k = k.view(n, spec.n_kv_heads, spec.head_dim).transpose(0, 1)
    v = v.view(n, spec.n_kv_heads, spec.head_dim).transpose(0, 1)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    qscore = None
    if prompt_len:                                          # query-based group (qwen25_lvu.py:663-664, 684-686; lvu_cache.py:99-116):
        ng = n - prompt_len                                 # the last prompt_len rows are the prompt; their K/V never enter the cache
        qscore = query_attention_scores(q[:, ng:], k[:, :ng].contiguous())
        k, v = k[:, :ng], v[:, :ng]
    k_all, v_all = cache.append(layer, k.contiguous(), v.contiguous())
    # n queries over kv = past + (n - prompt_len) keys, flash-attn bottom-right alignment: query i sees keys j <= i + kv - n
    att = attention_bottom_right(q, k_all, v_all, spec.head_dim ** -0.5)
    h = h + torch.nn.functional.linear(att.reshape(n, -1), w[p + "o_proj.weight"])
    kept = None
    if k_keep is not None:                                  # post_process_kv_cache, utils.py:257-342
        past = k_all.shape[1] - (n - prompt_len)            # utils.py:236-238: q_len -= prompt_length
        if prompt_len:
            if prune_hidden:
                raise NotImplementedError("query-based scoring + hidden-state pruning: the reference drops the prompt rows there")
            norms = query_score_keys(qscore, v_all[:, past:] if predict_type == "query_attention_weights_by_value_norm" else None)
            source, order = 0, 1
        else:
            source, order = NORM_PRUNE_MODES[predict_type]
            norms = _norm_keys_of((v_all if source else k_all)[:, past:])
        kept = select_k_largest(norms, k_keep) if order else select_k_smallest(norms, k_keep)
        ti = torch.from_numpy(kept.astype(np.int64))
        cache.k[layer] = torch.cat([k_all[:, :past], k_all[:, past:][:, ti]], dim=1)
        cache.v[layer] = torch.cat([v_all[:, :past], v_all[:, past:][:, ti]], dim=1)
        if trace is not None:
            trace.setdefault("norm_bits", []).append(norms)
        if prune_hidden:
            h, cos, sin = h[ti], cos[ti], sin[ti]
    x = rmsnorm(h, w[p + "post_attention_layernorm.weight"], spec.rms_eps)
    gate = torch.nn.functional.linear(x, w[p + "mlp.gate_proj.weight"])
    up = torch.nn.functional.linear(x, w[p + "mlp.up_proj.weight"])
    h = h + torch.nn.functional.linear(torch.nn.functional.silu(gate) * up, w[p + "mlp.down_proj.weight"])
    return h, kept, cos, sin


def group_prefill(w: dict, spec: TextSpec, embeds: torch.Tensor, pos: np.ndarray, group_tokens: Sequence[int],
                  cfg: PruneCfg, want_logits: bool = True):
    """Whole path: group loop (qwen25_lvu.py:671-717) + prompt tail without pruning (:724-742).

    embeds [T, d] already holds text embeddings with the video rows overwritten by the ViT output
    (masked_scatter [3P]); pos int64 [3, T].  Returns dict(logits [V] of the last position — the
    first generated token's distribution —, kept[g][l] index lists, cache_len[l], cache).
    """
    dt = embeds.dtype
    cache = OracleCache(spec.n_layers)
    kept_all: List[List[Optional[np.ndarray]]] = []
    start = 0
    post = torch.from_numpy(pos)
    segments = list(group_tokens) + [embeds.shape[0] - sum(group_tokens)]
    query_based = "query" in cfg.top_k_predict_type          # lvu_config.py:31-33
    m = segments[-1] if query_based else 0                   # prompt = everything after the last video token (qwen25_lvu.py:661)
    h = None
    for gi, n in enumerate(segments):
        is_tail = gi == len(segments) - 1
        pl = 0 if is_tail else m
        h = embeds[start:start + n]
        if pl:                                               # group tokens + the prompt tokens; positions = the next n+m of the sequence (:684-689)
            h = torch.cat([h, embeds[-m:]], 0)
        cos, sin = mrope_cos_sin(post[:, start:start + n + pl], spec, dt)
        kept_g = []
        for l in range(spec.n_layers):
            q_len = h.shape[0] - pl
            k_keep = None if is_tail else effective_k(q_len, cfg.top_k, cfg.top_p, cfg.top_k_decay_type,
                                                      cfg.top_k_decay_factor, l, spec.n_layers, cfg.enable,
                                                      cfg.top_k_starting_layer)
            ph = (not is_tail and cfg.enable and isinstance(cfg.prefill_prune_starting_layer, int)
                  and cfg.prefill_prune_starting_layer >= 0 and l >= cfg.prefill_prune_starting_layer)
            h, kept, cos, sin = decoder_layer(h, w, l, spec, cache, cos, sin, k_keep, prune_hidden=ph,
                                              predict_type=cfg.top_k_predict_type, prompt_len=pl)
            kept_g.append(kept)
        kept_all.append(kept_g)
        start += n
    out = {"kept": kept_all, "cache_len": [cache.length(l) for l in range(spec.n_layers)], "cache": cache}
    if want_logits:
        x = rmsnorm(h[-1:], w["norm.weight"], spec.rms_eps)
        out["logits"] = torch.nn.functional.linear(x, w["lm_head.weight"])[0].float()
    return out


# --------------------------------------------------------------------------------------
# Deterministic synthetic weights (numpy RandomState: frozen stream) — shared recipe with
# quickvideo_amd/weights.py, restated here so the oracle stays import-free of the product.
# --------------------------------------------------------------------------------------

def synthetic_text_weights(spec: TextSpec, seed: int = 0, dtype: torch.dtype = torch.float32, std: float = 0.02,
                           bias_std: float = 0.02, norm_jitter: float = 0.0) -> dict:
    rs = np.random.RandomState(seed)

    def mat(*shape, s=std):
        return torch.from_numpy((rs.standard_normal(shape) * s).astype(np.float32)).to(dtype)

    w = {"embed_tokens.weight": mat(spec.vocab, spec.hidden)}
    qd, kd = spec.n_heads * spec.head_dim, spec.n_kv_heads * spec.head_dim
    for l in range(spec.n_layers):
        p = f"layers.{l}."
        w[p + "input_layernorm.weight"] = (torch.ones(spec.hidden) + mat(spec.hidden, s=norm_jitter).float()).to(dtype)
        w[p + "q_proj.weight"] = mat(qd, spec.hidden); w[p + "q_proj.bias"] = mat(qd, s=bias_std)
        w[p + "k_proj.weight"] = mat(kd, spec.hidden); w[p + "k_proj.bias"] = mat(kd, s=bias_std)
        w[p + "v_proj.weight"] = mat(kd, spec.hidden); w[p + "v_proj.bias"] = mat(kd, s=bias_std)
        w[p + "o_proj.weight"] = mat(spec.hidden, qd)
        w[p + "post_attention_layernorm.weight"] = (torch.ones(spec.hidden) + mat(spec.hidden, s=norm_jitter).float()).to(dtype)
        w[p + "mlp.gate_proj.weight"] = mat(spec.intermediate, spec.hidden)
        w[p + "mlp.up_proj.weight"] = mat(spec.intermediate, spec.hidden)
        w[p + "mlp.down_proj.weight"] = mat(spec.hidden, spec.intermediate)
    w["norm.weight"] = (torch.ones(spec.hidden) + mat(spec.hidden, s=norm_jitter).float()).to(dtype)
    w["lm_head.weight"] = w["embed_tokens.weight"] if spec.tie_embeddings else mat(spec.vocab, spec.hidden)
    return w


# --------------------------------------------------------------------------------------
# Hash-generated weights: bit-identical on every device (integer hash -> exact float steps),
# so a fixture made from them on the build container's CPU can be replayed on the GPU box
# without shipping 15 GB of weights.  Used by the full-depth 7B-dim parity case (GV8).
# --------------------------------------------------------------------------------------

def _lowbias32(x: torch.Tensor) -> torch.Tensor:
    """32-bit integer mixer on int64 tensors (all intermediates < 2^63: exact on CPU and GPU)."""
    m = 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    return x ^ (x >> 16)


def hashed_normal(shape, seed: int, std: float, device="cpu", dtype=torch.bfloat16, chunk: int = 1 << 25) -> torch.Tensor:
    """Approximately N(0, std^2) (Irwin-Hall: sum of 3 hashed uniforms, |x| <= 3 std) as a pure function of
    (seed, flat index): element i is the same bit pattern whichever device computes it (integer ops are exact,
    every float step is a single IEEE operation, bf16 cast = RNE)."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=dtype, device=device)
    base = (int(seed) * 0x9E3779B1) & 0xFFFFFFFF
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        i = torch.arange(c0, c1, dtype=torch.int64, device=device) * 3 + base
        acc = None
        for r in range(3):
            u = (_lowbias32((i + r) & 0xFFFFFFFF) >> 8).to(torch.float32) * (1.0 / 16777216.0)
            acc = u if acc is None else acc + u
        out[c0:c1] = ((acc - 1.5) * (2.0 * std)).to(dtype)
    return out.view(*shape)


def hashed_text_weights(spec: TextSpec, seed: int = 0, device="cpu", dtype=torch.bfloat16, std: float = 0.02, bias_std: float = 0.02,
                        norm_jitter: float = 0.1, layers: Optional[Sequence[int]] = None, with_embed: bool = True) -> dict:
    """HF-named decoder weights from hashed_normal; every tensor has its own seed (seed, layer, slot), so any subset of
    layers can be generated alone and agrees with the full model."""
    d, qd, kd, I = spec.hidden, spec.n_heads * spec.head_dim, spec.n_kv_heads * spec.head_dim, spec.intermediate
    hn = lambda shape, s, sd: hashed_normal(shape, s, sd, device, dtype)
    one = lambda s: (1.0 + hashed_normal((d,), s, norm_jitter, device, torch.float32)).to(dtype)
    w = {}
    for l in (range(spec.n_layers) if layers is None else layers):
        p, s0 = f"layers.{l}.", seed * 100_003 + (l + 1) * 101
        w[p + "input_layernorm.weight"] = one(s0 + 0)
        w[p + "q_proj.weight"], w[p + "q_proj.bias"] = hn((qd, d), s0 + 1, std), hn((qd,), s0 + 2, bias_std)
        w[p + "k_proj.weight"], w[p + "k_proj.bias"] = hn((kd, d), s0 + 3, std), hn((kd,), s0 + 4, bias_std)
        w[p + "v_proj.weight"], w[p + "v_proj.bias"] = hn((kd, d), s0 + 5, std), hn((kd,), s0 + 6, bias_std)
        w[p + "o_proj.weight"] = hn((d, qd), s0 + 7, std)
        w[p + "post_attention_layernorm.weight"] = one(s0 + 8)
        w[p + "mlp.gate_proj.weight"] = hn((I, d), s0 + 9, std)
        w[p + "mlp.up_proj.weight"] = hn((I, d), s0 + 10, std)
        w[p + "mlp.down_proj.weight"] = hn((d, I), s0 + 11, std)
    w["norm.weight"] = one(seed * 100_003 + 7)
    if with_embed:
        w["embed_tokens.weight"] = hn((spec.vocab, d), seed * 100_003 + 11, std)
        w["lm_head.weight"] = w["embed_tokens.weight"] if spec.tie_embeddings else hn((spec.vocab, d), seed * 100_003 + 13, std)
    return w


def hashed_state_dict(names_shapes, seed: int, device="cpu", dtype=torch.bfloat16, std: float = 0.03, norm_jitter: float = 0.05) -> dict:
    """name -> hashed_normal tensor for a list of (name, shape); 1-D tensors whose name says "norm" / "ln_q" get 1 + jitter (weights) or
    plain jitter (biases).  Used for the vision-tower fixtures (GV10): the same bits wherever they are generated."""
    out = {}
    for i, (name, shape) in enumerate(names_shapes):
        is_norm = ("norm" in name or "ln_q" in name) and len(shape) == 1
        t = hashed_normal(tuple(shape), seed * 7919 + i, norm_jitter if is_norm else std, device, torch.float32)
        if is_norm and name.endswith("weight"):
            t = t + 1.0
        out[name] = t.to(dtype)
    return out
