"""Model geometry of the Qwen2-VL / Qwen2.5-VL LLM decoder the hot path runs."""
from dataclasses import dataclass, replace
from typing import Tuple


@dataclass(frozen=True)
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
    # special token ids (HF Qwen2-VL config)
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    # temporal M-RoPE step per temporal patch.  Qwen2-VL: 1.0.  Qwen2.5-VL: second_per_grid_t * tokens_per_second with
    # second_per_grid_t = temporal_patch_size / sampled fps (get_rope_index [3P]) — stored as MINUS tokens_per_second and resolved per
    # video by resolved_temporal_scale(); a positive value is used as is.
    temporal_scale: float = 1.0

    def resolved_temporal_scale(self, sample_fps: float, temporal_patch_size: int = 2) -> float:
        if self.temporal_scale >= 0:
            return self.temporal_scale
        return -self.temporal_scale * temporal_patch_size / sample_fps

    @property
    def q_dim(self) -> int:
        return self.n_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.n_kv_heads * self.head_dim

    def linear_flops_per_token(self) -> float:
        d, L = self.hidden, self.n_layers
        return 2.0 * L * (d * self.q_dim + 2 * d * self.kv_dim + self.q_dim * d + 3 * d * self.intermediate)

    def attn_flops(self, n: int, prefix: int) -> float:
        """4*L*Hq*D*(n*P + n(n+1)/2)  (SURVEY.md §8d)."""
        return 4.0 * self.n_layers * self.n_heads * self.head_dim * (n * prefix + n * (n + 1) / 2.0)


QWEN2_VL_2B = TextSpec(hidden=1536, n_heads=12, n_kv_heads=2, head_dim=128, intermediate=8960, n_layers=28, vocab=151936,
                       tie_embeddings=True)
QWEN2_VL_7B = TextSpec(hidden=3584, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, n_layers=28, vocab=152064)
QWEN2_VL_72B = TextSpec(hidden=8192, n_heads=64, n_kv_heads=8, head_dim=128, intermediate=29568, n_layers=80, vocab=152064)
TINY = TextSpec(hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, intermediate=512, n_layers=3, vocab=320,
                video_token_id=300, vision_start_token_id=301, vision_end_token_id=302)

PRESETS = {"qwen2-vl-2b": QWEN2_VL_2B, "qwen2-vl-7b": QWEN2_VL_7B, "qwen2-vl-72b": QWEN2_VL_72B, "tiny": TINY,
           "qwen2.5-vl-7b": replace(QWEN2_VL_7B, temporal_scale=-2.0)}           # tokens_per_second = 2: 2.0 per grid step only at 2 fps
