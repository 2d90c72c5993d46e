"""Pins quickvideo_amd/vit.py (pure torch, device-agnostic) against the installed transformers Qwen2-VL vision
tower with the same weights, and the GPU-side patchify against the HF processor's layout rule.  CPU, fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from quickvideo_amd.vit import VisionSpec, VisionTower, VisionWeights, patchify_frames, vision_pos_ids, CLIP_MEAN, CLIP_STD


def hf_tower(spec):
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLVisionConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
    cfg = Qwen2VLVisionConfig(depth=spec.depth, embed_dim=spec.embed_dim, hidden_size=spec.out_hidden, num_heads=spec.num_heads,
                              mlp_ratio=int(spec.mlp_ratio), patch_size=spec.patch_size, spatial_merge_size=spec.spatial_merge_size,
                              temporal_patch_size=spec.temporal_patch_size, hidden_act="quick_gelu")
    cfg._attn_implementation = "eager"
    torch.manual_seed(3)
    m = Qwen2VisionTransformerPretrainedModel(cfg).eval().float()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.05)
    return m


@pytest.mark.parametrize("grid", [(2, 4, 6), (1, 8, 4), (3, 2, 2)])
def test_vit_matches_hf(grid):
    spec = VisionSpec(depth=2, embed_dim=64, num_heads=4, mlp_ratio=2.0, out_hidden=96)
    m = hf_tower(spec)
    w = VisionWeights.from_named(spec, m.state_dict(), "cpu", dtype=torch.float32)
    t, h, wd = grid
    rs = np.random.RandomState(0)
    pix = torch.from_numpy(rs.standard_normal((t * h * wd, spec.patch_dim)).astype(np.float32))
    with torch.no_grad():
        ref = m(pix, grid_thw=torch.tensor([list(grid)])).pooler_output
    got = VisionTower(w).forward(pix, grid)
    assert got.shape == ref.shape == (t * h * wd // 4, 96)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4), (got - ref).abs().max()


def test_patchify_layout():
    """Same (t, h/2, w/2, 2, 2 | C, T, 14, 14) order and CLIP normalisation as HF's Qwen2VLImageProcessor."""
    spec = VisionSpec(depth=1, embed_dim=16, num_heads=2)
    rs = np.random.RandomState(1)
    F_, H, W = 4, 56, 84
    frames = torch.from_numpy(rs.randint(0, 256, (F_, 3, H, W)).astype(np.uint8))
    rows, grid = patchify_frames(frames, spec, dtype=torch.float32)
    assert grid == (2, 4, 6) and rows.shape == (2 * 4 * 6, 1176)
    x = frames.numpy().astype(np.float64) / 255.0
    x = (x - np.array(CLIP_MEAN).reshape(1, 3, 1, 1)) / np.array(CLIP_STD).reshape(1, 3, 1, 1)
    # reference rule written independently: patch index -> (t, hb, wb, hi, wi)
    for idx in (0, 1, 5, 23, 24, 47):
        t, rem = divmod(idx, 24)
        hb, rem = divmod(rem, 12)          # 12 = (w/2 blocks = 3) * 4
        wb, rem = divmod(rem, 4)
        hi, wi = divmod(rem, 2)
        ph, pw = hb * 2 + hi, wb * 2 + wi
        patch = x[t * 2:(t + 1) * 2, :, ph * 14:(ph + 1) * 14, pw * 14:(pw + 1) * 14]     # [T, C, 14, 14]
        want = patch.transpose(1, 0, 2, 3).reshape(-1)                                   # (C, T, 14, 14)
        assert np.allclose(rows[idx].numpy(), want, atol=1e-5)
    pos = vision_pos_ids(grid, 2, "cpu")
    assert pos[:6].tolist() == [[0, 0], [0, 1], [1, 0], [1, 1], [0, 2], [0, 3]]


@pytest.mark.parametrize("grid", [(2, 8, 12), (1, 10, 6), (3, 4, 4), (1, 18, 22)])
def test_qwen25_tower_matches_hf(grid):
    """Qwen2.5-VL tower (the reference's model family): RMSNorm, gated-SiLU MLP, windowed attention + 4 full layers."""
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VisionTransformerPretrainedModel
    spec = VisionSpec(arch="qwen2.5", depth=3, embed_dim=64, num_heads=4, out_hidden=96, intermediate=80, window_size=112, fullatt_blocks=(1,))
    cfg = Qwen2_5_VLVisionConfig(depth=3, hidden_size=64, intermediate_size=80, num_heads=4, out_hidden_size=96, window_size=112,
                                 fullatt_block_indexes=[1], patch_size=14, spatial_merge_size=2, temporal_patch_size=2, hidden_act="silu")
    cfg._attn_implementation = "eager"
    torch.manual_seed(5)
    m = Qwen2_5_VisionTransformerPretrainedModel(cfg).eval().float()
    for p in m.parameters():
        torch.nn.init.normal_(p, std=0.05)
    w = VisionWeights.from_named(spec, m.state_dict(), "cpu", dtype=torch.float32)
    t, h, wd = grid
    pix = torch.from_numpy(np.random.RandomState(1).standard_normal((t * h * wd, spec.patch_dim)).astype(np.float32))
    with torch.no_grad():
        ref = m(pix, grid_thw=torch.tensor([list(grid)])).pooler_output
    got = VisionTower(w).forward(pix, grid)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, atol=3e-5, rtol=1e-4), (got - ref).abs().max()


@pytest.mark.parametrize("name", ["qwen2", "qwen25"])
def test_towers_match_hf_fixture_at_real_width(name, golden_dir):
    """GV10 (made by oracle/make_golden.py from the installed transformers towers at width 1280 / head_dim 80): the product's torch
    tower on the CPU in fp32 from the same hash-generated bf16 weights.  The GPU test pins the HIP tower to the same file."""
    import json, os
    from oracle import qp_oracle as O
    meta = json.load(open(os.path.join(golden_dir, "gv10_vit_towers.json")))[name]
    ref = np.load(os.path.join(golden_dir, "gv10_vit_towers.npz"))[f"{name}_out"]
    kw = dict(meta["spec"]); kw["fullatt_blocks"] = tuple(kw.get("fullatt_blocks", ()))
    spec = VisionSpec(arch=meta["arch"], **{k: v for k, v in kw.items() if k != "fullatt_blocks" or meta["arch"] == "qwen2.5"})
    sd = O.hashed_state_dict([(n, tuple(s)) for n, s in meta["names_shapes"]], meta["weight_seed"])
    w = VisionWeights.from_named(spec, {k: v.float() for k, v in sd.items()}, "cpu", dtype=torch.float32)
    t, h, wd = meta["grid"]
    pix = O.hashed_normal((t * h * wd, 1176), meta["pixel_seed"], 1.0).float()
    got = VisionTower(w).forward(pix, tuple(meta["grid"])).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-3 * meta["out_absmax"], np.abs(got - ref).max()


def test_qwen25_padded_mlp_is_the_unpadded_mlp():
    """vit.VisionTower._qwen25_mlp_padded: the intermediate width of the Qwen2.5-VL vision MLP (3420: no multiple of 8, so no 16-byte
    vectors, and of no GEMM tile) zero-padded to a multiple of 128.  Padded gate / up columns are silu(0) * 0 = 0 and meet zero columns of
    the padded down projection: in exact arithmetic — and in fp32 here — the padded MLP IS the unpadded one."""
    from quickvideo_amd.vit import VisionBlockWeights25, VisionTower
    g = torch.Generator().manual_seed(3)
    d, it, n = 64, 172, 37
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64) * 0.3
    b = VisionBlockWeights25(None, None, None, None, None, None, r(it, d), r(it), r(it, d), r(it), r(d, it), r(d))
    y = r(n, d)
    want = F.linear(F.silu(F.linear(y, b.gate_w, b.gate_b)) * F.linear(y, b.up_w, b.up_b), b.down_w, b.down_b)
    gu_w, gu_b, down_wp = VisionTower._qwen25_mlp_padded(b)
    ip = gu_w.shape[0] // 2
    assert ip == 256 and down_wp.shape == (d, ip) and VisionTower._qwen25_mlp_padded(b)[0] is gu_w      # built once
    gu = F.linear(y, gu_w, gu_b)
    got = F.linear(F.silu(gu[:, :ip]) * gu[:, ip:], down_wp, b.down_b)
    assert torch.equal(gu[:, it:ip], torch.zeros(n, ip - it, dtype=torch.float64)) and torch.allclose(got, want, rtol=0, atol=1e-12)
