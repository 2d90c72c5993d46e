"""LVUConfig / LVULayerConfig — field names, defaults and per-layer flags of the reference kept drop-in
(lvu/lvu_config.py:3-55)."""
from dataclasses import dataclass
from typing import Optional


_IGNORED_WARNED: set = set()


def _warn_ignored_once(key: str, msg: str):
    if key not in _IGNORED_WARNED:
        _IGNORED_WARNED.add(key)
        import warnings
        warnings.warn(msg, stacklevel=4)


@dataclass
class LVUConfig:
    model_name_or_path: str
    model_type: str = "qwen2vl_mi355x"        # reference default "qwen25_vl"; ours is the native plugin
    top_k_predict_type: str = "key_norms_small"
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    top_k_starting_layer: Optional[int] = None
    do_top_k_for_query: bool = False
    adaptive_local_attention: bool = True
    video_group_size: Optional[int] = None     # per frame
    prefill_prune_starting_layer: Optional[int] = None
    fps: Optional[int] = None
    num_frames: int = 32
    use_tqdm: bool = False
    extra_kwargs: Optional[dict] = None
    enable: bool = True
    cache_dir: Optional[str] = None
    save_video_cache: bool = False
    top_k_decay_factor: Optional[float] = None
    top_k_decay_type: Optional[str] = None
    query_based: bool = False

    def __post_init__(self):
        # lvu_config.py:26-33
        if self.top_k_decay_type == "linear" and self.top_k_decay_factor is None:
            print(f"Warning: top_k_decay_type is set to {self.top_k_decay_type} but top_k_decay_factor is None. Setting it to 0.5.")
            self.top_k_decay_factor = 0.5
        if "query" in self.top_k_predict_type:
            self.query_based = True
        # The reference's frame cache (cache_dir / save_video_cache: JPEG-encoded frames + processor outputs on disk, qwen25_lvu.py:552-592,
        # lvu_cache.py:28-49 — and a NameError in the interleaved plugin) is outside this package's scope (§8: the hot
        # path starts at decoded frames).  The fields stay for drop-in construction; setting them does nothing HERE, and says so once.
        if self.save_video_cache or self.cache_dir:
            _warn_ignored_once("save_video_cache / cache_dir", "LVUConfig.save_video_cache / cache_dir are accepted for drop-in compatibility but the "
                               "on-disk frame cache is not implemented by the MI355X-native plugin: every generate() reads the frames from the video source")


@dataclass
class LVULayerConfig:
    layer_idx: int
    total_layers: int
    lvu_config: LVUConfig
    is_last_layer: bool = False
    prune_for_next_layer: bool = False

    def __post_init__(self):
        # lvu_config.py:42-55
        if self.layer_idx is None:
            raise ValueError("layer_idx cannot be None")
        self.is_last_layer = (self.layer_idx == self.total_layers - 1)
        p = self.lvu_config.prefill_prune_starting_layer
        self.prune_for_next_layer = bool(isinstance(p, int) and p >= 0 and self.layer_idx >= p)


# norm-based predict types (utils.py:117-136): name -> (norm source: 0 keys / 1 values, order: 0 k smallest / 1 k largest)
NORM_PRUNE_MODES = {"key_norms_small": (0, 0), "key_norms": (0, 1), "vector_norms_small": (1, 0), "vector_norms": (1, 1)}


# query-based predict types (utils.py:55-62; lvu_config.py:31-33 sets query_based): name -> weight the score by the value norm?
QUERY_PRUNE_MODES = {"query_attention_weights": False, "query_attention_weights_by_value_norm": True}


def effective_k(q_len: int, cfg: LVUConfig, layer_idx: int, total_layers: int) -> Optional[int]:
    """How many of the group's q_len new tokens this layer keeps; None = this layer does not prune.

    lvu/utils.py:231-255.  int(q_len * top_p) is an IEEE-double multiply then truncation.
    top_k_starting_layer: the reference reads a non-existent field there (AttributeError, utils.py:253);
    we implement the evident intent (layers below it do not prune).
    """
    top_k, top_p = cfg.top_k, cfg.top_p
    if top_p is not None and top_p >= 0:
        top_k = min((top_k or q_len), int(q_len * top_p))
    if not cfg.top_k_decay_type:
        pass
    elif cfg.top_k_decay_type == "linear":
        top_k = top_k - int(top_k * (layer_idx / total_layers))
    elif cfg.top_k_decay_type == "exponential":
        top_k = int(top_k * (cfg.top_k_decay_factor ** layer_idx))
    else:
        raise ValueError(f"Unknown top_k_decay_type: {cfg.top_k_decay_type}")
    if not cfg.enable or not top_k or top_k <= 0 or q_len <= top_k:
        return None
    s = cfg.top_k_starting_layer
    if isinstance(s, int) and s > 0 and layer_idx < s:
        return None
    return int(top_k)
