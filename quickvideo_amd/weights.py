"""Decoder weights resident in HBM, laid out for the fused projections of the engine.

q/k/v projections are concatenated into one [ (Hq+2Hkv)*D, d ] matrix (+bias) and gate/up into one
[2I, d] matrix, so a layer is four GEMMs.  Under tensor parallelism each rank holds its head slice of
qkv / o_proj and its column slice of gate_up / down (quickvideo_amd/tp.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .spec import TextSpec


@dataclass
class LayerWeights:
    ln1: torch.Tensor        # [d]
    w_qkv: torch.Tensor      # [(hq+2hkv)*D, d]   rows: q heads, k heads, v heads (local heads under TP)
    b_qkv: torch.Tensor      # [(hq+2hkv)*D]
    w_o: torch.Tensor        # [d, hq*D]
    ln2: torch.Tensor        # [d]
    w_gate_up: torch.Tensor  # [2*I, d]           rows: gate then up (local columns under TP)
    w_down: torch.Tensor     # [d, I]


@dataclass
class DecoderWeights:
    spec: TextSpec
    embed: torch.Tensor      # [V, d]
    layers: List[LayerWeights]
    norm: torch.Tensor       # [d]
    lm_head: torch.Tensor    # [V, d]
    tp_rank: int = 0
    tp_size: int = 1

    @property
    def local_q_heads(self) -> int:
        return self.layers[0].w_o.shape[1] // self.spec.head_dim

    @property
    def local_kv_heads(self) -> int:
        return (self.layers[0].w_qkv.shape[0] // self.spec.head_dim - self.local_q_heads) // 2

    @property
    def local_inter(self) -> int:
        return self.layers[0].w_down.shape[1]

    @staticmethod
    def from_named(spec: TextSpec, sd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16, tp_rank: int = 0,
                   tp_size: int = 1) -> "DecoderWeights":
        """Build from HF-style names ("layers.{i}.q_proj.weight" or "layers.{i}.self_attn.q_proj.weight", ...)."""
        def g(*names):
            for nm in names:
                if nm in sd:
                    return sd[nm]
            raise KeyError(names[0])

        D, hq, hkv, I = spec.head_dim, spec.n_heads, spec.n_kv_heads, spec.intermediate
        assert hq % tp_size == 0 and I % tp_size == 0, "tensor-parallel degree must divide heads and intermediate size"
        # KV heads: shard when divisible, replicate otherwise (7B has 4 KV heads: TP=8 replicates each on 2 ranks)
        lq = hq // tp_size
        q_lo = tp_rank * lq
        if hkv % tp_size == 0:
            lkv = hkv // tp_size; kv_lo = tp_rank * lkv
        else:
            assert tp_size % hkv == 0, "n_kv_heads and tp_size must divide one another"
            lkv = 1; kv_lo = tp_rank // (tp_size // hkv)
        li = I // tp_size; i_lo = tp_rank * li
        to = lambda t: t.to(device=device, dtype=dtype).contiguous()
        layers = []
        for l in range(spec.n_layers):
            p = f"layers.{l}."
            a = lambda s: (p + s, p + "self_attn." + s)
            qw, kw, vw = g(*a("q_proj.weight")), g(*a("k_proj.weight")), g(*a("v_proj.weight"))
            qb, kb, vb = g(*a("q_proj.bias")), g(*a("k_proj.bias")), g(*a("v_proj.bias"))
            ow = g(*a("o_proj.weight"))
            w_qkv = torch.cat([qw[q_lo * D:(q_lo + lq) * D], kw[kv_lo * D:(kv_lo + lkv) * D], vw[kv_lo * D:(kv_lo + lkv) * D]], 0)
            b_qkv = torch.cat([qb[q_lo * D:(q_lo + lq) * D], kb[kv_lo * D:(kv_lo + lkv) * D], vb[kv_lo * D:(kv_lo + lkv) * D]], 0)
            gw, uw, dw = g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"), g(p + "mlp.down_proj.weight")
            layers.append(LayerWeights(
                ln1=to(g(p + "input_layernorm.weight")), w_qkv=to(w_qkv), b_qkv=to(b_qkv),
                w_o=to(ow[:, q_lo * D:(q_lo + lq) * D]), ln2=to(g(p + "post_attention_layernorm.weight")),
                w_gate_up=to(torch.cat([gw[i_lo:i_lo + li], uw[i_lo:i_lo + li]], 0)), w_down=to(dw[:, i_lo:i_lo + li])))
        embed = to(g("embed_tokens.weight"))
        lm = embed if spec.tie_embeddings and "lm_head.weight" not in sd else to(g("lm_head.weight"))
        return DecoderWeights(spec, embed, layers, to(g("norm.weight")), lm, tp_rank, tp_size)

    @staticmethod
    def synthetic(spec: TextSpec, device, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02, tp_rank: int = 0,
                  tp_size: int = 1, n_layers: Optional[int] = None) -> "DecoderWeights":
        """Seeded random weights at the real dims, generated on the device (no checkpoint offline).
        Under TP every rank draws only its own shard (seeded by (seed, layer, rank))."""
        D, hq, hkv, I, d = spec.head_dim, spec.n_heads, spec.n_kv_heads, spec.intermediate, spec.hidden
        assert hq % tp_size == 0 and I % tp_size == 0
        lq = hq // tp_size
        lkv = hkv // tp_size if hkv % tp_size == 0 else 1
        li = I // tp_size
        gen = torch.Generator(device=device)

        def mat(*shape, s=std):
            return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * s).to(dtype)

        gen.manual_seed(seed)
        embed = mat(spec.vocab, d)
        lm_head = embed if spec.tie_embeddings else mat(spec.vocab, d)
        norm = torch.ones(d, device=device, dtype=dtype)
        layers = []
        for l in range(spec.n_layers if n_layers is None else n_layers):
            gen.manual_seed(seed * 1_000_003 + l * 1009 + tp_rank + 1)
            layers.append(LayerWeights(
                ln1=torch.ones(d, device=device, dtype=dtype), w_qkv=mat((lq + 2 * lkv) * D, d), b_qkv=mat((lq + 2 * lkv) * D),
                w_o=mat(d, lq * D), ln2=torch.ones(d, device=device, dtype=dtype), w_gate_up=mat(2 * li, d), w_down=mat(d, li)))
        return DecoderWeights(spec, embed, layers, norm, lm_head, tp_rank, tp_size)
