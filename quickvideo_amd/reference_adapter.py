"""Reference-side adapter: the two objects a maintainer of TIGER-AI-Lab/QuickVideo drops into `lvu/` to run the reference's OWN
patched forwards (lvu/models/qwen25_lvu.py:29-212) on libquickprefill.so — see INTEGRATION.md §1.

  * `ArenaLVUCache`   stands where `LVUCache(DynamicCache)` stands (lvu/lvu_cache.py:68-117): `update(key_states, value_states,
                      layer_idx, cache_kwargs)` appends IN PLACE into a pre-allocated `[Hkv, capacity, 128]` arena per layer and
                      returns views of rows [0, len) — no torch.cat of the whole past; plus the three accessors the native prune
                      needs: `layer(i)`, `set_len(i, n)`, `workspace(nbytes)`.
  * `post_process_kv_cache`  has the reference's 8-argument signature and 6-tuple result (lvu/utils.py:197-206, 376) and replaces
                      its body: effective-k in Python (pure integers), then ONE `qp_prune_tail` call on the arena (key-norm ->
                      select -> in-place compaction, no host sync), and `qp_gather_rows` for the hidden-state hand-off.

Everything goes through the C ABI (`quickvideo_amd.native.QuickPrefillOps` = the ctypes table of include/quickprefill.h)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch

from .lvu_config import LVULayerConfig, NORM_PRUNE_MODES, effective_k
from .native import QuickPrefillOps


class ArenaLVUCache:
    def __init__(self, n_layers: int, n_kv_heads: int, capacity: int, head_dim: int = 128, device="cuda", dtype=torch.bfloat16,
                 ops: Optional[QuickPrefillOps] = None):
        self.ops = ops if ops is not None else QuickPrefillOps(torch.device(device))
        self.capacity, self.head_dim, self.n_kv = capacity, head_dim, n_kv_heads
        self.buf = torch.empty(n_layers, 2, n_kv_heads, capacity, head_dim, dtype=dtype, device=device)
        self.len: List[int] = [0] * n_layers
        self.prompt_length = 0                      # query-based mode is served by the native engine, not by this adapter
        self._ws: Optional[torch.Tensor] = None

    # ---- the HF Cache surface the reference's patched attention uses (qwen25_lvu.py:56-58)
    def update(self, key_states: torch.Tensor, value_states: torch.Tensor, layer_idx: int,
               cache_kwargs: Optional[Dict[str, Any]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """key/value_states [1, Hkv, n, D] (post-RoPE) -> appended at rows [len, len+n); returns [1, Hkv, len+n, D] views."""
        assert key_states.shape[0] == 1, "Only support batch size 1 for now"         # utils.py:264
        n, past = key_states.shape[2], self.len[layer_idx]
        if past + n > self.capacity:
            raise ValueError(f"KV arena overflow: {past}+{n} > capacity {self.capacity}")
        self.buf[layer_idx, 0, :, past:past + n].copy_(key_states[0])
        self.buf[layer_idx, 1, :, past:past + n].copy_(value_states[0])
        self.len[layer_idx] = past + n
        return self.buf[layer_idx, 0, :, :past + n][None], self.buf[layer_idx, 1, :, :past + n][None]

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.len[layer_idx]

    def __len__(self):
        return len(self.len)

    def __getitem__(self, layer_idx: int):
        k, v, n = self.layer(layer_idx)
        return k[:, :n][None], v[:, :n][None]

    # ---- accessors for the native prune
    def layer(self, layer_idx: int) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """(K arena [Hkv, capacity, D], V arena, rows in use)."""
        return self.buf[layer_idx, 0], self.buf[layer_idx, 1], self.len[layer_idx]

    def set_len(self, layer_idx: int, n: int):
        self.len[layer_idx] = n

    def workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.buf.device)
        return self._ws


def post_process_kv_cache(hidden_states: torch.Tensor, attention_mask: torch.Tensor = None, position_ids: torch.Tensor = None,
                          cache_position: torch.Tensor = None, position_embeddings: Tuple[torch.Tensor, torch.Tensor] = None,
                          attn_weights: torch.Tensor = None, present_key_value: ArenaLVUCache = None,
                          lvu_layer_config: LVULayerConfig = None):
    """Same signature, call point (qwen25_lvu.py:183-192) and result tuple as lvu/utils.py:197-376, for an ArenaLVUCache."""
    unchanged = (hidden_states, attention_mask, position_ids, cache_position, position_embeddings, present_key_value)
    if lvu_layer_config is None:
        return unchanged
    cfg, layer = lvu_layer_config.lvu_config, lvu_layer_config.layer_idx
    if cfg.top_k_predict_type not in NORM_PRUNE_MODES:
        raise ValueError(f"Unknown predict type: {cfg.top_k_predict_type}")           # utils.py:189
    if not isinstance(present_key_value, ArenaLVUCache):
        raise ValueError(f"Unknown present_key_value type: {type(present_key_value)}")    # utils.py:262
    q_len = hidden_states.shape[1]
    k = effective_k(q_len, cfg, layer, lvu_layer_config.total_layers)                  # utils.py:231-255
    if k is None:
        return unchanged
    assert hidden_states.shape[0] == 1, f"Only support batch size 1 for now, but got {hidden_states.shape[0]}"    # utils.py:264
    ops = present_key_value.ops
    src, order = NORM_PRUNE_MODES[cfg.top_k_predict_type]
    kc, vc, total = present_key_value.layer(layer)
    past = total - q_len                                                               # utils.py:266-271: only the new rows are scored
    idx = torch.empty(k, dtype=torch.int32, device=kc.device)
    ws = present_key_value.workspace(ops.prune_workspace_bytes(q_len, k, kc.shape[0], kc.shape[2]))
    ops.prune_tail(kc, vc, kc.stride(0), past, q_len, k, kc.shape[0], kc.shape[2], idx, ws, mode=(src << 1) | order)
    present_key_value.set_len(layer, past + k)                                         # replaces key_cache[layer] = cat(...) (utils.py:333-340)
    if lvu_layer_config.prune_for_next_layer:                                          # utils.py:292-331, 344-372
        def rows(t2d):                                                                  # [q_len, C] -> [k, C] on the device
            out = torch.empty(k, t2d.shape[1], dtype=t2d.dtype, device=t2d.device)
            ops.gather_rows(t2d.contiguous(), idx, k, t2d.shape[1] * t2d.element_size(), out)
            return out
        il = idx.long()
        hidden_states = rows(hidden_states[0])[None]
        if cache_position is not None:
            cache_position = cache_position.index_select(0, il)
        if position_ids is not None:
            position_ids = position_ids.index_select(-1, il)
        if attention_mask is not None and attention_mask.dim() == 2:
            attention_mask = attention_mask.index_select(1, il)
        if position_embeddings is not None:
            position_embeddings = tuple(pe.index_select(-2, il) for pe in position_embeddings)
    return hidden_states, attention_mask, position_ids, cache_position, position_embeddings, present_key_value
