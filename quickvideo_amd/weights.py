"""Decoder weights resident in HBM, laid out for the fused projections of the engine.

q/k/v projections are concatenated into one [ (Hq+2Hkv)*D, d ] matrix (+bias) and gate/up into one
[2I, d] matrix, so a layer is four GEMMs.  Under tensor parallelism each rank holds its head slice of
qkv / o_proj and its column slice of gate_up / down (quickvideo_amd/tp.py)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from .spec import TextSpec


def pp_layer_split(n_layers: int, world: int, rank: int) -> Tuple[int, int]:
    """Layer-pipeline stage `rank` of `world`: contiguous layers [l0, l1); the first n_layers % world stages take one more."""
    base, extra = divmod(n_layers, world)
    l0 = rank * base + min(rank, extra)
    return l0, l0 + base + (1 if rank < extra else 0)


def tp_head_partition(hq: int, hkv: int, tp_rank: int, tp_size: int):
    """(q head indices of this rank — -1 marks a zero pad head —, first kv head, number of local kv heads).

    hkv % tp == 0: kv heads sharded, each rank takes the q heads of its kv heads.  Otherwise tp % hkv == 0: every kv
    head is replicated on rep = tp/hkv consecutive ranks and its `group` q heads are dealt out ceil(group/rep) per
    rank, padded with zero heads (Qwen2-VL-7B at TP=8: 28 q / 4 kv heads -> ranks get 4+3(+1 pad) q heads)."""
    group = hq // hkv
    if hkv % tp_size == 0:
        lkv = hkv // tp_size
        kv_lo = tp_rank * lkv
        return list(range(kv_lo * group, (kv_lo + lkv) * group)), kv_lo, lkv
    assert tp_size % hkv == 0, "n_kv_heads and tp_size must divide one another"
    rep = tp_size // hkv
    kvh, sub = tp_rank // rep, tp_rank % rep
    per = -(-group // rep)
    idx = [kvh * group + sub * per + i if sub * per + i < group else -1 for i in range(per)]
    return idx, kvh, 1


def padded_inter(li: int) -> int:
    """Width a rank's MLP column shard is stored at.  A shard width that is no multiple of 64 (72B: 29568 / 8 = 3696, / 4 = 7392) makes
    the down projection's K ragged: measured on hipBLASLt (tools/probe/probe_gemm_alignment.py, profiles/r6j_*): K = 3696 -> 3712 is 24-26 %
    faster at M = 960 / 2240.  Such shards are zero-padded to the next multiple of 128 (so gate | up has N a multiple of 256): the padded
    gate / up rows give silu(0) * 0 = 0 and meet zero columns of the down shard — the same numbers, bit for bit in exact arithmetic."""
    return li if li % 64 == 0 else (li + 127) // 128 * 128


def pad_mlp_shard(w_gate_up: torch.Tensor, w_down: torch.Tensor):
    """([2 li, d], [d, li]) -> ([2 lip, d], [d, lip]) with lip = padded_inter(li): gate rows, zeros, up rows, zeros | zero columns."""
    li = w_down.shape[1]
    lip = padded_inter(li)
    if lip == li:
        return w_gate_up, w_down
    gu = torch.zeros(2 * lip, w_gate_up.shape[1], dtype=w_gate_up.dtype, device=w_gate_up.device)
    gu[:li], gu[lip:lip + li] = w_gate_up[:li], w_gate_up[li:]
    dn = torch.zeros(w_down.shape[0], lip, dtype=w_down.dtype, device=w_down.device)
    dn[:, :li] = w_down
    return gu, dn


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
    layer0: int = 0          # global index of layers[0] (layer-pipeline stages hold a contiguous slice)
    n_layers_total: int = -1 # layers of the whole model (-1: len(layers))

    def __post_init__(self):
        if self.n_layers_total < 0:
            self.n_layers_total = len(self.layers)

    @property
    def local_q_heads(self) -> int:
        return self.layers[0].w_o.shape[1] // self.spec.head_dim

    @property
    def local_kv_heads(self) -> int:
        return (self.layers[0].w_qkv.shape[0] // self.spec.head_dim - self.local_q_heads) // 2

    @property
    def local_inter(self) -> int:
        """Columns of this rank's MLP shard AS STORED (zero-padded to a GEMM-friendly width when I / tp is ragged: padded_inter)."""
        return self.layers[0].w_down.shape[1]

    @staticmethod
    def from_named(spec: TextSpec, sd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16, tp_rank: int = 0,
                   tp_size: int = 1, layer_range: Optional[Tuple[int, int]] = None) -> "DecoderWeights":
        """Build from HF-style names ("layers.{i}.q_proj.weight" or "layers.{i}.self_attn.q_proj.weight", ...).
        layer_range = (l0, l1): only layers [l0, l1) (a layer-pipeline stage)."""
        def g(*names):
            for nm in names:
                if nm in sd:
                    return sd[nm]
            raise KeyError(names[0])

        D, hq, hkv, I = spec.head_dim, spec.n_heads, spec.n_kv_heads, spec.intermediate
        assert I % tp_size == 0, "tensor-parallel degree must divide the intermediate size"
        q_idx, kv_lo, lkv = tp_head_partition(hq, hkv, tp_rank, tp_size)
        li = I // tp_size; i_lo = tp_rank * li

        def q_rows(w):            # rows of a [hq*D, ...] matrix / vector for this rank's q heads (zero rows for pad heads)
            parts = [w[h * D:(h + 1) * D] if h >= 0 else torch.zeros_like(w[:D]) for h in q_idx]
            return torch.cat(parts, 0)

        def q_cols(w):            # columns of o_proj [d, hq*D]
            parts = [w[:, h * D:(h + 1) * D] if h >= 0 else torch.zeros_like(w[:, :D]) for h in q_idx]
            return torch.cat(parts, 1)
        to = lambda t: t.to(device=device, dtype=dtype).contiguous()
        layers = []
        l0, l1 = layer_range if layer_range is not None else (0, spec.n_layers)
        for l in range(l0, l1):
            p = f"layers.{l}."
            a = lambda s: (p + s, p + "self_attn." + s)
            qw, kw, vw = g(*a("q_proj.weight")), g(*a("k_proj.weight")), g(*a("v_proj.weight"))
            qb, kb, vb = g(*a("q_proj.bias")), g(*a("k_proj.bias")), g(*a("v_proj.bias"))
            ow = g(*a("o_proj.weight"))
            w_qkv = torch.cat([q_rows(qw), kw[kv_lo * D:(kv_lo + lkv) * D], vw[kv_lo * D:(kv_lo + lkv) * D]], 0)
            b_qkv = torch.cat([q_rows(qb), kb[kv_lo * D:(kv_lo + lkv) * D], vb[kv_lo * D:(kv_lo + lkv) * D]], 0)
            gw, uw, dw = g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"), g(p + "mlp.down_proj.weight")
            w_gu, w_dn = to(torch.cat([gw[i_lo:i_lo + li], uw[i_lo:i_lo + li]], 0)), to(dw[:, i_lo:i_lo + li])
            if tp_size > 1:
                w_gu, w_dn = pad_mlp_shard(w_gu, w_dn)
            layers.append(LayerWeights(
                ln1=to(g(p + "input_layernorm.weight")), w_qkv=to(w_qkv), b_qkv=to(b_qkv),
                w_o=to(q_cols(ow)), ln2=to(g(p + "post_attention_layernorm.weight")), w_gate_up=w_gu, w_down=w_dn))
        embed = to(g("embed_tokens.weight"))
        lm = embed if spec.tie_embeddings and "lm_head.weight" not in sd else to(g("lm_head.weight"))
        return DecoderWeights(spec, embed, layers, to(g("norm.weight")), lm, tp_rank, tp_size, l0, spec.n_layers)

    @staticmethod
    def synthetic(spec: TextSpec, device, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02, tp_rank: int = 0,
                  tp_size: int = 1, n_layers: Optional[int] = None, layer_range: Optional[Tuple[int, int]] = None) -> "DecoderWeights":
        """Seeded random weights at the real dims, generated on the device (no checkpoint offline).
        Every layer has its own seed and is drawn in FULL (the single-GPU tensors), then cut down to this rank's tensor-parallel
        shard with the same head / column partition as `from_named` — so every TP degree, every layer-pipeline stage and the
        single-process model are the SAME model (a TP run must reproduce the single-GPU first token)."""
        D, hq, hkv, I, d = spec.head_dim, spec.n_heads, spec.n_kv_heads, spec.intermediate, spec.hidden
        assert I % tp_size == 0
        q_idx, kv_lo, lkv = tp_head_partition(hq, hkv, tp_rank, tp_size)
        li = I // tp_size
        i_lo = tp_rank * li
        gen = torch.Generator(device=device)

        def mat(*shape, s=std):
            return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * s).to(dtype)

        def q_rows(w):            # rows of a [hq*D, ...] matrix / vector for this rank's q heads (zero rows for pad heads)
            return torch.cat([w[h * D:(h + 1) * D] if h >= 0 else torch.zeros_like(w[:D]) for h in q_idx], 0)

        def q_cols(w):            # columns of o_proj [d, hq*D]
            return torch.cat([w[:, h * D:(h + 1) * D] if h >= 0 else torch.zeros_like(w[:, :D]) for h in q_idx], 1)

        gen.manual_seed(seed)
        embed = mat(spec.vocab, d)
        lm_head = embed if spec.tie_embeddings else mat(spec.vocab, d)
        norm = torch.ones(d, device=device, dtype=dtype)
        layers = []
        total = spec.n_layers if n_layers is None else n_layers
        l0, l1 = layer_range if layer_range is not None else (0, total)
        for l in range(l0, l1):
            gen.manual_seed(seed * 1_000_003 + l * 1009 + 1)
            w_qkv, b_qkv, w_o, w_gu, w_dn = mat((hq + 2 * hkv) * D, d), mat((hq + 2 * hkv) * D), mat(d, hq * D), mat(2 * I, d), mat(d, I)
            if tp_size > 1:
                kq, kk = hq * D, (hq + hkv) * D
                ksl, vsl = slice(kq + kv_lo * D, kq + (kv_lo + lkv) * D), slice(kk + kv_lo * D, kk + (kv_lo + lkv) * D)
                w_qkv = torch.cat([q_rows(w_qkv[:kq]), w_qkv[ksl], w_qkv[vsl]], 0).contiguous()
                b_qkv = torch.cat([q_rows(b_qkv[:kq]), b_qkv[ksl], b_qkv[vsl]], 0).contiguous()
                w_o = q_cols(w_o).contiguous()
                w_gu = torch.cat([w_gu[i_lo:i_lo + li], w_gu[I + i_lo:I + i_lo + li]], 0).contiguous()
                w_dn = w_dn[:, i_lo:i_lo + li].contiguous()
                w_gu, w_dn = pad_mlp_shard(w_gu, w_dn)
            layers.append(LayerWeights(ln1=torch.ones(d, device=device, dtype=dtype), w_qkv=w_qkv, b_qkv=b_qkv, w_o=w_o,
                                       ln2=torch.ones(d, device=device, dtype=dtype), w_gate_up=w_gu, w_down=w_dn))
        return DecoderWeights(spec, embed, layers, norm, lm_head, tp_rank, tp_size, l0, total)
