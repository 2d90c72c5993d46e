"""Qwen2-VL vision front end on PyTorch-ROCm (north_star: "ViT patch-embed/attention front
end ... rebuilt on PyTorch-ROCm").

What the reference does on the CPU (HF processor under the GIL, qwen25_lvu_interleaved.py:252-271, 318-340:
rescale + CLIP-normalise + patchify the fp32 frames, then H2D of 4n x 1176 fp32) happens here on the GPU from the
uint8 frames: upload 1 byte/pixel instead of 4, one fused normalise+patchify pass, then the ViT
(transformers Qwen2VisionTransformerPretrainedModel [3P], restated): Conv3d patch embed as a GEMM, `depth`
pre-LN blocks with full attention inside each temporal patch (cu_seqlens = one sequence per grid_t), 2-D rotary
position embedding on (h, w), quick-GELU MLP, and the 2x2 PatchMerger MLP into the LLM width.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass(frozen=True)
class VisionSpec:
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_ratio: float = 4.0
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    in_channels: int = 3
    out_hidden: int = 3584          # LLM width
    # Qwen2.5-VL tower (the reference's model family, lvu.py:60): RMSNorm, gated-SiLU MLP, windowed attention except in
    # `fullatt_blocks`; arch "qwen2" = Qwen2-VL tower (LayerNorm, quick-GELU MLP, full attention per temporal patch).
    arch: str = "qwen2"
    intermediate: int = 3420
    window_size: int = 112
    fullatt_blocks: Tuple[int, ...] = (7, 15, 23, 31)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size

    def flops_per_patch(self) -> float:
        d, m = self.embed_dim, int(self.embed_dim * self.mlp_ratio)
        return 2.0 * (self.patch_dim * d + self.depth * (4 * d * d + 2 * d * m)) + 2.0 * (4 * d * 4 * d + 4 * d * self.out_hidden) / 4


QWEN25_VL_VIT_7B = VisionSpec(arch="qwen2.5", out_hidden=3584)
QWEN2_VL_VIT_7B = VisionSpec(out_hidden=3584)
QWEN2_VL_VIT_2B = VisionSpec(out_hidden=1536)
QWEN2_VL_VIT_72B = VisionSpec(out_hidden=8192)
TINY_VIT = VisionSpec(depth=2, embed_dim=64, num_heads=4, mlp_ratio=2.0, out_hidden=256)


_CLIP_CONST: dict = {}


def _clip_constants(device):
    """CLIP mean/std as device tensors, built once per device: a torch.tensor(..., device=cuda) per group is a blocking pageable
    H2D copy that stalls the launch thread (and used to hide a ring-slot race in the pipeline by accident)."""
    key = str(device)
    if key not in _CLIP_CONST:
        _CLIP_CONST[key] = (torch.tensor(CLIP_MEAN, dtype=torch.float32).view(1, 3, 1, 1).to(device),
                            torch.tensor(CLIP_STD, dtype=torch.float32).view(1, 3, 1, 1).to(device))
    return _CLIP_CONST[key]


def patchify_frames(frames_u8: torch.Tensor, spec: VisionSpec, dtype=torch.bfloat16) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
    """uint8 frames [F, 3, H, W] (already at the smart_resize target size) -> (pixel rows [grid_t*grid_h*grid_w, 1176],
    (grid_t, grid_h, grid_w)) in the HF Qwen2VLImageProcessor order (t, h/2, w/2, 2, 2 | C, T, 14, 14) [3P]."""
    Fn, C, H, W = frames_u8.shape
    ps, tp, mg = spec.patch_size, spec.temporal_patch_size, spec.spatial_merge_size
    assert Fn % tp == 0 and H % (ps * mg) == 0 and W % (ps * mg) == 0, "frame count / size not aligned to the patch grid"
    gt, gh, gw = Fn // tp, H // ps, W // ps
    mean, std = _clip_constants(frames_u8.device)
    x = (frames_u8.to(torch.float32) * (1.0 / 255.0) - mean) / std
    x = x.view(gt, tp, C, gh // mg, mg, ps, gw // mg, mg, ps).permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
    return x.reshape(gt * gh * gw, C * tp * ps * ps).to(dtype), (gt, gh, gw)


def vision_pos_ids(grid_thw: Tuple[int, int, int], merge: int, device) -> torch.Tensor:
    """(h, w) position of every patch in the merge-grouped order [3P get_vision_position_ids / rot_pos_emb]."""
    t, h, w = grid_thw
    hp = torch.arange(h, device=device).unsqueeze(1).expand(-1, w)
    wp = torch.arange(w, device=device).unsqueeze(0).expand(h, -1)
    re = lambda p: p.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
    return torch.stack([re(hp), re(wp)], dim=-1).repeat(t, 1)


@dataclass
class VisionBlockWeights:
    ln1_w: torch.Tensor; ln1_b: torch.Tensor
    qkv_w: torch.Tensor; qkv_b: torch.Tensor
    proj_w: torch.Tensor; proj_b: torch.Tensor
    ln2_w: torch.Tensor; ln2_b: torch.Tensor
    fc1_w: torch.Tensor; fc1_b: torch.Tensor
    fc2_w: torch.Tensor; fc2_b: torch.Tensor


@dataclass
class VisionBlockWeights25:
    n1: torch.Tensor; n2: torch.Tensor                       # RMSNorm weights
    qkv_w: torch.Tensor; qkv_b: torch.Tensor
    proj_w: torch.Tensor; proj_b: torch.Tensor
    gate_w: torch.Tensor; gate_b: torch.Tensor
    up_w: torch.Tensor; up_b: torch.Tensor
    down_w: torch.Tensor; down_b: torch.Tensor


@dataclass
class VisionWeights:
    spec: VisionSpec
    patch_w: torch.Tensor                 # [embed_dim, patch_dim]  (Conv3d weight flattened)
    blocks: List[VisionBlockWeights]
    ln_q_w: torch.Tensor; ln_q_b: torch.Tensor
    m1_w: torch.Tensor; m1_b: torch.Tensor
    m2_w: torch.Tensor; m2_b: torch.Tensor

    @staticmethod
    def from_named(spec: VisionSpec, sd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16, prefix: str = "") -> "VisionWeights":
        g = lambda k: sd[prefix + k].to(device=device, dtype=dtype).contiguous()
        blocks = []
        if spec.arch == "qwen2.5":
            for i in range(spec.depth):
                p = f"blocks.{i}."
                blocks.append(VisionBlockWeights25(g(p + "norm1.weight"), g(p + "norm2.weight"), g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias"),
                                                   g(p + "attn.proj.weight"), g(p + "attn.proj.bias"), g(p + "mlp.gate_proj.weight"),
                                                   g(p + "mlp.gate_proj.bias"), g(p + "mlp.up_proj.weight"), g(p + "mlp.up_proj.bias"),
                                                   g(p + "mlp.down_proj.weight"), g(p + "mlp.down_proj.bias")))
            return VisionWeights(spec, g("patch_embed.proj.weight").reshape(spec.embed_dim, -1).contiguous(), blocks,
                                 g("merger.ln_q.weight"), None, g("merger.mlp.0.weight"), g("merger.mlp.0.bias"),
                                 g("merger.mlp.2.weight"), g("merger.mlp.2.bias"))
        for i in range(spec.depth):
            p = f"blocks.{i}."
            blocks.append(VisionBlockWeights(g(p + "norm1.weight"), g(p + "norm1.bias"), g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias"),
                                             g(p + "attn.proj.weight"), g(p + "attn.proj.bias"), g(p + "norm2.weight"), g(p + "norm2.bias"),
                                             g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"), g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias")))
        return VisionWeights(spec, g("patch_embed.proj.weight").reshape(spec.embed_dim, -1).contiguous(), blocks,
                             g("merger.ln_q.weight"), g("merger.ln_q.bias"), g("merger.mlp.0.weight"), g("merger.mlp.0.bias"),
                             g("merger.mlp.2.weight"), g("merger.mlp.2.bias"))

    @staticmethod
    def synthetic(spec: VisionSpec, device, seed: int = 0, dtype=torch.bfloat16, std: float = 0.02) -> "VisionWeights":
        gen = torch.Generator(device=device); gen.manual_seed(seed + 77)
        mat = lambda *s: (torch.randn(*s, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)
        one = lambda n: torch.ones(n, device=device, dtype=dtype)
        zero = lambda n: torch.zeros(n, device=device, dtype=dtype)
        d, m, d4 = spec.embed_dim, int(spec.embed_dim * spec.mlp_ratio), spec.embed_dim * spec.spatial_merge_size ** 2
        if spec.arch == "qwen2.5":
            it = spec.intermediate
            blocks = [VisionBlockWeights25(one(d), one(d), mat(3 * d, d), mat(3 * d), mat(d, d), mat(d), mat(it, d), mat(it), mat(it, d), mat(it),
                                           mat(d, it), mat(d)) for _ in range(spec.depth)]
            return VisionWeights(spec, mat(d, spec.patch_dim), blocks, one(d), None, mat(d4, d4), mat(d4), mat(spec.out_hidden, d4),
                                 mat(spec.out_hidden))
        blocks = [VisionBlockWeights(one(d), zero(d), mat(3 * d, d), mat(3 * d), mat(d, d), mat(d), one(d), zero(d), mat(m, d), mat(m),
                                     mat(d, m), mat(d)) for _ in range(spec.depth)]
        return VisionWeights(spec, mat(d, spec.patch_dim), blocks, one(d), zero(d), mat(d4, d4), mat(d4), mat(spec.out_hidden, d4),
                             mat(spec.out_hidden))


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def b_contig(w) -> bool:
    return all(blk.fc1_w.is_contiguous() and blk.fc2_w.is_contiguous() for blk in w.blocks)


class VisionTower:
    """Stateless forward over VisionWeights; runs on whatever device the weights live on (MI355X in the product)."""

    def __init__(self, weights: VisionWeights, ops=None):
        """ops: quickvideo_amd.native.QuickPrefillOps -> rotary / attention / quick-GELU run as HIP kernels
        (qp_vit_rope, qp_vit_attn, qp_quick_gelu; head_dim 80 only); None -> the same math in plain torch ops."""
        self.w, self.spec = weights, weights.spec
        self.ops = ops if (ops is not None and weights.spec.head_dim == 80 and hasattr(ops, "vit_attn")) else None
        self._lt_shapes: dict = {}             # (name, rows) -> True once hipBLASLt's candidates were timed for this GEMM shape

    def _lt(self, name: str, x: torch.Tensor, w: torch.Tensor, bias, act: int = 0, alpha: float = 1.0, peers=None) -> torch.Tensor:
        """out = act(alpha x w^T + bias) through the library's hipBLASLt path with the candidate that is FASTEST for this shape: the first
        time a (projection, row count) shows up, qp_linear_tune times every heuristic candidate over the same projection of all blocks
        (`peers`, cold weights) and keeps the best — torch's F.linear takes the library's first candidate, which for these K = 1280 shapes
        is up to 1.3x slower (tools/bench_vit.py: QP_VIT_LT=0 / 1).  Also one call that keeps the interpreter lock instead of three torch
        calls that drop it (bench.py host_contention)."""
        x = x if x.is_contiguous() else x.contiguous()
        out = torch.empty(x.shape[0], w.shape[0], dtype=x.dtype, device=x.device)
        key = (name, x.shape[0], act, bias is not None)
        if key not in self._lt_shapes:
            self.ops.linear_tune(x, [pw for pw in (peers or [w])], bias, out, act, alpha)
            self._lt_shapes[key] = True
        self.ops.linear_act(x, w, bias, out, act, alpha)
        return out

    # ------------------------------------------------------------------ front end, first step: frames -> pixel rows
    def _patch_lut(self, device) -> torch.Tensor:
        """[3, 256] bf16: what patchify_frames makes of byte v in channel c — the SAME torch expression on the 256 byte values (so the HIP
        gather below is bit-identical to the torch path by construction)."""
        luts = self.__dict__.setdefault("_luts", {})
        key = str(device)
        if key not in luts:
            mean, std = _clip_constants(device)
            v = torch.arange(256, device=device).to(torch.uint8).view(1, 1, 1, 256).expand(1, 3, 1, 256)
            luts[key] = ((v.to(torch.float32) * (1.0 / 255.0) - mean) / std).to(torch.bfloat16).reshape(3, 256).contiguous()
        return luts[key]

    def _patch_w(self, k: int) -> torch.Tensor:
        """The patch-embedding weight for pixel rows of k columns: the checkpoint's [embed, 1176], or — rows from the HIP patchify, padded
        with zero columns to a GEMM-friendly K — the same weight with zero columns appended (built once)."""
        w = self.w.patch_w
        if k == w.shape[1]:
            return w
        cache = self.__dict__.setdefault("_patch_w_padded", {})
        if k not in cache:
            assert k > w.shape[1], (k, tuple(w.shape))
            wp = torch.zeros(w.shape[0], k, dtype=w.dtype, device=w.device)
            wp[:, :w.shape[1]] = w
            cache[k] = wp
        return cache[k]

    def patchify(self, frames_u8: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
        """patchify_frames for this tower.  On the GPU with the library: ONE kernel (qp_patchify: uint8 -> normalised bf16 rows in the HF patch
        order through a 3 x 256 table of the torch path's own values; columns padded with zeros to a multiple of 128 for the patch-embedding
        GEMM) instead of five torch passes; anywhere else the torch function.  QP_VIT_HIP_PATCHIFY=0: torch (A/B)."""
        s, ops = self.spec, self.ops
        if (ops is not None and hasattr(ops, "patchify") and frames_u8.is_cuda and frames_u8.dtype == torch.uint8
                and self.w.patch_w.dtype == torch.bfloat16 and os.environ.get("QP_VIT_HIP_PATCHIFY", "1") == "1"):
            Fn, C, H, W = frames_u8.shape
            ps, tp, mg = s.patch_size, s.temporal_patch_size, s.spatial_merge_size
            assert C == s.in_channels == 3 and Fn % tp == 0 and H % (ps * mg) == 0 and W % (ps * mg) == 0, "frame count / size not aligned to the patch grid"
            gt, gh, gw = Fn // tp, H // ps, W // ps
            out = torch.empty(gt * gh * gw, (s.patch_dim + 127) // 128 * 128, dtype=torch.bfloat16, device=frames_u8.device)
            ops.patchify(frames_u8.contiguous(), ps, tp, mg, self._patch_lut(frames_u8.device), out)
            return out, (gt, gh, gw)
        return patchify_frames(frames_u8, s, self.w.patch_w.dtype)

    @torch.no_grad()
    def forward(self, pixel_rows: torch.Tensor, grid_thw: Tuple[int, int, int]) -> torch.Tensor:
        if self.spec.arch == "qwen2.5":
            return self._forward_qwen25(pixel_rows, grid_thw)
        s, w = self.spec, self.w
        t, h, wd = grid_thw
        n, seq = pixel_rows.shape[0], h * wd
        assert n == t * seq
        x = F.linear(pixel_rows.to(w.patch_w.dtype), self._patch_w(pixel_rows.shape[1]))     # Conv3d(stride = kernel) == GEMM
        pos = vision_pos_ids(grid_thw, s.spatial_merge_size, x.device)                  # [n, 2]
        rd = s.head_dim // 2
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rd, 2, dtype=torch.float32, device=x.device) / rd))
        rot = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)                         # [n, head_dim/2]
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]                         # fp32 [n, 1, head_dim]
        H, hd = s.num_heads, s.head_dim
        ops = self.ops
        if ops is not None:
            cos_h, sin_h = rot.cos().contiguous(), rot.sin().contiguous()                # fp32 [n, head_dim/2]
        fused_ln = (ops is not None and hasattr(ops, "add_layernorm") and s.embed_dim % 8 == 0 and s.embed_dim <= 4096
                    and os.environ.get("QP_VIT_FUSED_LN", "1") == "1")               # developer A/B switch (tools/bench_vit.py)
        pend = None                                   # residual branch not yet added to x (fused into the next LayerNorm launch)
        ybuf = torch.empty_like(x) if fused_ln else None
        fused_act = (ops is not None and hasattr(ops, "linear_act") and os.environ.get("QP_VIT_FUSED_ACT", "1") == "1"
                     and x.is_cuda and b_contig(w))
        # every GEMM of the tower through the tuned hipBLASLt path (QP_VIT_LT=0: torch's F.linear / addmm, the round-3 form, for A/B)
        use_lt = fused_act and hasattr(ops, "linear_tune") and os.environ.get("QP_VIT_LT", "1") == "1"
        blocks = w.blocks

        def norm(wt, bs):
            """x += pending residual; LayerNorm(x)  (Qwen2VLVisionBlock: x = x + attn(norm1(x)); x = x + mlp(norm2(x)))"""
            nonlocal x, pend
            if fused_ln:
                ops.add_layernorm(x, pend, wt, bs, ybuf, 1e-6)                           # one pass: residual add + LayerNorm
                pend = None
                return ybuf
            if pend is not None:
                x, pend = x + pend, None
            return F.layer_norm(x, (s.embed_dim,), wt, bs, 1e-6)

        if fused_ln:
            x = x.contiguous()
        # all blocks in ONE library call (qp_vit_blocks: the same launches, sequenced inside the library) once the GEMM plans of this row
        # count exist — the first pass at a new row count goes block by block below and tunes them.  QP_VIT_ONE_CALL=0: always block by block.
        one_call = (use_lt and fused_ln and hasattr(ops, "vit_blocks") and os.environ.get("QP_VIT_ONE_CALL", "1") == "1"
                    and all(((nm, n, act, True) in self._lt_shapes) for nm, act in (("qkv", 0), ("proj", 0), ("fc1", ops.ACT_SWISH), ("fc2", 0))))
        if one_call:
            from .native import QpVitBlock
            arr = self.__dict__.get("_blocks_arr")
            if arr is None:
                arr = (QpVitBlock * len(blocks))()
                for i, b in enumerate(blocks):
                    if getattr(b, "_fc1_b_scaled", None) is None:
                        b._fc1_b_scaled = (b.fc1_b.float() * 1.702).contiguous()
                    for f, tns in (("ln1_w", b.ln1_w), ("ln1_b", b.ln1_b), ("qkv_w", b.qkv_w), ("qkv_b", b.qkv_b), ("proj_w", b.proj_w), ("proj_b", b.proj_b),
                                   ("ln2_w", b.ln2_w), ("ln2_b", b.ln2_b), ("fc1_w", b.fc1_w), ("fc1_bias_scaled", b._fc1_b_scaled), ("fc2_w", b.fc2_w),
                                   ("fc2_b", b.fc2_b)):
                        setattr(arr[i], f, tns.data_ptr())
                self._blocks_arr = arr
            e = lambda *shape: torch.empty(*shape, dtype=x.dtype, device=x.device)
            mlp = blocks[0].fc1_w.shape[0]
            qkv_b, att_b, pend, z_b = e(n, 3 * s.embed_dim), e(n, s.embed_dim), e(n, s.embed_dim), e(n, mlp)
            ops.vit_blocks(arr, len(blocks), t, seq, s.embed_dim, H, mlp, x, ybuf, qkv_b, att_b, pend, z_b, cos_h, sin_h, 1e-6)
        for b in (() if one_call else w.blocks):
            y = norm(b.ln1_w, b.ln1_b)
            qkv = (self._lt("qkv", y, b.qkv_w, b.qkv_b, peers=[bb.qkv_w for bb in blocks]) if use_lt
                   else F.linear(y, b.qkv_w, b.qkv_b))                                   # [n, 3*H*hd]
            if ops is not None:
                ops.vit_rope(qkv, cos_h, sin_h, H, hd)                                   # q, k rotated in place
                a = torch.empty(n, H * hd, dtype=x.dtype, device=x.device)
                ops.vit_attn(qkv, t, seq, H, hd, hd ** -0.5, a)                          # one sequence per temporal patch
            else:
                qkv = qkv.view(n, 3, H, hd)
                q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
                qf, kf = q.float(), k.float()
                q = (qf * cos + _rotate_half(qf) * sin).to(x.dtype)
                k = (kf * cos + _rotate_half(kf) * sin).to(x.dtype)
                q4, k4, v4 = (z.reshape(t, seq, H, hd).transpose(1, 2) for z in (q, k, v))   # [t, H, seq, hd]
                a = F.scaled_dot_product_attention(q4, k4, v4, is_causal=False)
                a = a.transpose(1, 2).reshape(n, H * hd)
            pend = self._lt("proj", a, b.proj_w, b.proj_b, peers=[bb.proj_w for bb in blocks]) if use_lt else F.linear(a, b.proj_w, b.proj_b)
            y = norm(b.ln2_w, b.ln2_b)
            if use_lt:
                if getattr(b, "_fc1_b_scaled", None) is None:
                    b._fc1_b_scaled = (b.fc1_b.float() * 1.702).contiguous()
                z = self._lt("fc1", y, b.fc1_w, b._fc1_b_scaled, act=ops.ACT_SWISH, alpha=1.702, peers=[bb.fc1_w for bb in blocks])
                pend = self._lt("fc2", z, b.fc2_w, b.fc2_b, alpha=1.0 / 1.702, peers=[bb.fc2_w for bb in blocks])
                continue
            if fused_act:
                # fc1 + quick-GELU in ONE GEMM: Swish epilogue on 1.702 (y W1^T + b1) = 1.702 quick_gelu(.), the 1/1.702 rides on fc2's alpha
                if getattr(b, "_fc1_b_scaled", None) is None:
                    b._fc1_b_scaled = (b.fc1_b.float() * 1.702).contiguous()
                z = torch.empty(n, b.fc1_w.shape[0], dtype=x.dtype, device=x.device)
                ops.linear_act(y.contiguous(), b.fc1_w, b._fc1_b_scaled, z, ops.ACT_SWISH, alpha=1.702)
                pend = torch.addmm(b.fc2_b, z, b.fc2_w.t(), alpha=1.0 / 1.702)
                continue
            y = F.linear(y, b.fc1_w, b.fc1_b)
            if ops is not None:
                ops.quick_gelu(y, y)
            else:
                y = y * torch.sigmoid(1.702 * y)                                         # quick_gelu
            pend = F.linear(y, b.fc2_w, b.fc2_b)
        x = norm(w.ln_q_w, w.ln_q_b)
        y = x.view(-1, s.embed_dim * s.spatial_merge_size ** 2)
        y = F.gelu(F.linear(y, w.m1_w, w.m1_b))
        return F.linear(y, w.m2_w, w.m2_b)                                               # [n/4, out_hidden]


    # ------------------------------------------------------------------ Qwen2.5-VL tower
    @staticmethod
    def window_index(grid_thw: Tuple[int, int, int], merge: int, window_size: int, patch_size: int, device):
        """transformers get_vision_window_index [3P]: permutation of the merged (2x2) token units into window-major order
        and the cumulative patch counts of the windows (ragged at the borders)."""
        t, h, w = grid_thw
        vw = window_size // merge // patch_size
        lh, lw = h // merge, w // merge
        idx = torch.arange(t * lh * lw).reshape(t, lh, lw)
        ph, pw = vw - lh % vw, vw - lw % vw
        nh, nw = (lh + ph) // vw, (lw + pw) // vw
        pad = F.pad(idx, (0, pw, 0, ph), "constant", -100).reshape(t, nh, vw, nw, vw).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, vw, vw)
        seqlens = (pad != -100).sum([2, 3]).reshape(-1)
        flat = pad.reshape(-1)
        win = flat[flat != -100]
        lens = seqlens[seqlens > 0] * merge * merge
        return win.to(device), lens.to(device)

    @staticmethod
    def _qwen25_mlp_padded(b):
        """(gate | up weight [2 Ip, d], bias [2 Ip], down weight [d, Ip]) of a Qwen2.5-VL vision block with the intermediate width zero-
        padded to a multiple of 128; built once per block and kept beside the checkpoint's tensors."""
        cached = getattr(b, "_mlp_padded", None)
        if cached is None:
            it, d = b.gate_w.shape
            ip = (it + 127) // 128 * 128
            gu_w = torch.zeros(2 * ip, d, dtype=b.gate_w.dtype, device=b.gate_w.device)
            gu_b = torch.zeros(2 * ip, dtype=b.gate_b.dtype, device=b.gate_b.device)
            gu_w[:it], gu_w[ip:ip + it] = b.gate_w, b.up_w
            gu_b[:it], gu_b[ip:ip + it] = b.gate_b, b.up_b
            down_wp = torch.zeros(b.down_w.shape[0], ip, dtype=b.down_w.dtype, device=b.down_w.device)
            down_wp[:, :it] = b.down_w
            cached = b._mlp_padded = (gu_w, gu_b, down_wp)
        return cached

    def _forward_qwen25(self, pixel_rows: torch.Tensor, grid_thw: Tuple[int, int, int]) -> torch.Tensor:
        s, w = self.spec, self.w
        t, h, wd = grid_thw
        n, seq, unit = pixel_rows.shape[0], h * wd, s.spatial_merge_size ** 2
        H, hd, d = s.num_heads, s.head_dim, s.embed_dim
        x = F.linear(pixel_rows.to(w.patch_w.dtype), self._patch_w(pixel_rows.shape[1]))
        # the window permutation depends only on the grid: built once per (grid, device) — it is host work + an H2D copy
        wk = (tuple(grid_thw), str(x.device))
        cache = self.__dict__.setdefault("_win_cache", {})
        if wk not in cache:
            win_, lens_ = self.window_index(grid_thw, s.spatial_merge_size, s.window_size, s.patch_size, x.device)
            cu_ = torch.zeros(lens_.numel() + 1, dtype=torch.int32, device=x.device)
            cu_[1:] = torch.cumsum(lens_, 0).to(torch.int32)
            cache[wk] = (win_, lens_, cu_, int(lens_.max()), torch.argsort(win_))
        win, lens, cu_win, lmax_win, win_inv = cache[wk]
        x = x.reshape(n // unit, unit, d)[win].reshape(n, d)                             # window-major token order
        pos = vision_pos_ids(grid_thw, s.spatial_merge_size, x.device)
        rd = hd // 2
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rd, 2, dtype=torch.float32, device=x.device) / rd))
        rot = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)
        rot = rot.reshape(n // unit, unit, -1)[win].reshape(n, -1)
        emb = torch.cat((rot, rot), dim=-1)
        cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
        ops = self.ops
        if ops is not None:
            cos_h, sin_h = rot.cos().contiguous(), rot.sin().contiguous()
        # windows padded to the longest one: gather map [n_win, Lmax] (-1 = pad) for the windowed layers
        nwin, lmax = lens.numel(), int(lens.max())
        starts = torch.cumsum(lens, 0) - lens
        ar = torch.arange(lmax, device=x.device)
        wmap = starts[:, None] + ar[None, :]
        wvalid = ar[None, :] < lens[:, None]
        wmap = torch.where(wvalid, wmap, torch.zeros_like(wmap))
        rms_t = lambda z, g: (z.float() * torch.rsqrt(z.float().pow(2).mean(-1, keepdim=True) + 1e-6)).to(z.dtype) * g
        fused = ops is not None and d % 8 == 0
        fused_mlp = fused and hasattr(ops, "swiglu") and x.is_cuda and os.environ.get("QP_VIT25_FUSED_MLP", "1") == "1"   # A/B: tools/bench_vit.py
        pend = None                                   # residual branch not yet added to x (fused into the next RMSNorm launch)
        ybuf = torch.empty(n, d, dtype=x.dtype, device=x.device) if fused else None
        if fused:
            x = x.contiguous()

        def rms(g):
            """x += pending residual; RMSNorm(x) * g   (one qp_add_rmsnorm launch on the GPU)"""
            nonlocal x, pend
            if fused:
                ops.add_rmsnorm(x, pend, g, ybuf, 1e-6)
                pend = None
                return ybuf
            if pend is not None:
                x, pend = x + pend, None
            return rms_t(x, g)

        # QP_VIT25_LT=1 (opt-in, A/B): the tower's GEMMs through the library's hipBLASLt path with the candidate qp_linear_tune measured fastest
        # (over the same projection of all blocks, cold weights) instead of torch's F.linear (hipBLASLt's first candidate).  Measured
        # (profiles/r6n_qwen25_tower_tuned_gemms_ab.txt): a group of the 1-hour video 15.4-15.5 -> 15.1 ms (-2 %), a 560x1008 group 34.1-34.5 ->
        # 35.6 ms (+3.5 %: at 23 040 rows the stopwatch's picks lose to the default) — not a default.
        blocks = w.blocks
        use_lt = fused_mlp and hasattr(ops, "linear_tune") and os.environ.get("QP_VIT25_LT", "0") == "1"
        if use_lt:
            padded = [self._qwen25_mlp_padded(bb) for bb in blocks]
        for li, b in enumerate(w.blocks):
            y = rms(b.n1)
            qkv = self._lt("qkv25", y, b.qkv_w, b.qkv_b, peers=[bb.qkv_w for bb in blocks]) if use_lt else F.linear(y, b.qkv_w, b.qkv_b)
            full = li in s.fullatt_blocks
            if ops is not None:
                ops.vit_rope(qkv, cos_h, sin_h, H, hd)
            varlen = ops is not None and hasattr(ops, "vit_attn_varlen") and os.environ.get("QP_VIT_WINDOW_HIP", "1") == "1"
            if ops is not None and full:
                a = torch.empty(n, H * hd, dtype=x.dtype, device=x.device)
                ops.vit_attn(qkv, t, seq, H, hd, hd ** -0.5, a)
            elif varlen:                                                               # 28 of 32 blocks: ragged windows, same MFMA kernel
                a = torch.empty(n, H * hd, dtype=x.dtype, device=x.device)
                ops.vit_attn_varlen(qkv, cu_win, lmax_win, H, hd, hd ** -0.5, a)
            else:
                q3 = qkv.view(n, 3, H, hd)
                q, k, v = q3[:, 0], q3[:, 1], q3[:, 2]
                if ops is None:
                    qf, kf = q.float(), k.float()
                    q = (qf * cos + _rotate_half(qf) * sin).to(x.dtype)
                    k = (kf * cos + _rotate_half(kf) * sin).to(x.dtype)
                if full:
                    q4, k4, v4 = (z.reshape(t, seq, H, hd).transpose(1, 2) for z in (q, k, v))
                    a = F.scaled_dot_product_attention(q4, k4, v4).transpose(1, 2).reshape(n, H * hd)
                else:                                                                  # ragged windows, padded + key mask
                    q4, k4, v4 = (z[wmap].transpose(1, 2) for z in (q, k, v))          # [n_win, H, Lmax, hd]
                    o = F.scaled_dot_product_attention(q4, k4, v4, attn_mask=wvalid[:, None, None, :])
                    a = torch.empty(n, H, hd, dtype=x.dtype, device=x.device)
                    a[wmap[wvalid]] = o.transpose(1, 2)[wvalid]
                    a = a.reshape(n, H * hd)
            pend = self._lt("proj25", a, b.proj_w, b.proj_b, peers=[bb.proj_w for bb in blocks]) if use_lt else F.linear(a, b.proj_w, b.proj_b)
            y = rms(b.n2)
            if use_lt:
                gu_w, gu_b, down_wp = padded[li]
                gu = self._lt("gu25", y, gu_w, gu_b, peers=[pp[0] for pp in padded])
                act = torch.empty(n, gu_w.shape[0] // 2, dtype=x.dtype, device=x.device)
                ops.swiglu(gu, act)
                pend = self._lt("down25", act, down_wp, b.down_b, peers=[pp[2] for pp in padded])
                continue
            if fused_mlp:
                # gate | up as ONE GEMM into [n, 2 Ip] and silu(gate) * up as ONE pass (qp_swiglu: torch's bf16 rounding points), instead of
                # two GEMMs + silu + mul (three passes over [n, I]).  I = 3420 is not a multiple of 8 (16-byte vectors) nor of any GEMM
                # tile: the weights are zero-padded ONCE to Ip = 3456 = 27 x 128 — padded gate / up columns are silu(0) * 0 = 0 and meet
                # zero columns of the padded down projection, so the result is that of the unpadded MLP.
                gu_w, gu_b, down_wp = self._qwen25_mlp_padded(b)
                gu = F.linear(y, gu_w, gu_b)
                act = torch.empty(n, gu_w.shape[0] // 2, dtype=x.dtype, device=x.device)
                ops.swiglu(gu, act)
                pend = F.linear(act, down_wp, b.down_b)
                continue
            y = F.silu(F.linear(y, b.gate_w, b.gate_b)) * F.linear(y, b.up_w, b.up_b)
            pend = F.linear(y, b.down_w, b.down_b)
        y = rms(w.ln_q_w).view(-1, d * unit)
        y = F.linear(F.gelu(F.linear(y, w.m1_w, w.m1_b)), w.m2_w, w.m2_b)
        return y[win_inv]                                                              # back to raster (t, h/2, w/2) order
