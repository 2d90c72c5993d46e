import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from quickvideo_amd.native import QuickPrefillOps
from quickvideo_amd.vit import VisionSpec, VisionTower, VisionWeights
ops = QuickPrefillOps(torch.device("cuda:0"))
rs = np.random.RandomState(2)
H, hd = 16, 80
for (t, S) in ((3, 200), (2, 1120), (1, 4), (3, 140), (1, 64), (1, 65)):
    n = t * S
    qkv = torch.from_numpy(rs.standard_normal((n, 3 * H * hd)).astype(np.float32)).to(torch.bfloat16).cuda()
    out = torch.empty(n, H * hd, dtype=torch.bfloat16, device="cuda")
    ops.vit_attn(qkv, t, S, H, hd, hd ** -0.5, out)
    q4, k4, v4 = (qkv.view(t, S, 3, H, hd)[:, :, i].transpose(1, 2).float() for i in range(3))
    ref = torch.softmax(q4 @ k4.transpose(-1, -2) * hd ** -0.5, -1) @ v4
    got = out.view(t, S, H, hd).transpose(1, 2).float()
    err = (got - ref).abs()
    print("attn", t, S, "maxerr", err.max().item(), "nan", torch.isnan(got).sum().item())
    if err.max() > 0.05:
        bad = (err > 0.05).nonzero()
        print(" first bad idx (t,h,s,d):", bad[:5].tolist(), "count", len(bad))
spec = VisionSpec(depth=2, embed_dim=1280, num_heads=16, mlp_ratio=4.0, out_hidden=256)
w = VisionWeights.synthetic(spec, "cuda:0", seed=4, std=0.03)
for grid in ((3, 10, 14), (2, 28, 40), (1, 2, 2)):
    n = grid[0] * grid[1] * grid[2]
    rows = torch.from_numpy(rs.standard_normal((n, spec.patch_dim)).astype(np.float32)).to(torch.bfloat16).cuda()
    ref = VisionTower(w).forward(rows, grid).float()
    got = VisionTower(w, ops=ops).forward(rows, grid).float()
    print("tower", grid, (got - ref).abs().max().item(), ref.abs().max().item())
