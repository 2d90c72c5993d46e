"""Micro-benchmark of the single-query decode attention (qp_decode_attn) over the cache length."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
D, hq, hkv = 128, 28, 4
for L in (1, 64, 1024, 11534, 50000, 200000):
    cap = L + 8
    g = torch.Generator(device="cuda"); g.manual_seed(L)
    q = torch.randn(hq, D, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(hkv, cap, D, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(hkv, cap, D, generator=g, device="cuda").to(torch.bfloat16)
    state = torch.tensor([L - 1, 0], dtype=torch.int64, device="cuda")
    out = torch.empty(hq, D, dtype=torch.bfloat16, device="cuda")
    ws = ops.decode_attn_workspace(hq, hkv)
    f = lambda: ops.decode_attn(q, k, v, cap * D, state, hq, hkv, D, D ** -0.5, out, ws)
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(200): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 200 * 1e3
    print(f"L={L}: {us:.1f} us per call (attn + combine), KV {2 * hkv * L * D * 2 / 1e6:.1f} MB -> {2 * hkv * L * D * 2 / us / 1e6:.2f} TB/s", flush=True)
