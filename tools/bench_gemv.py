"""GEMV (qp_gemv) micro-benchmark: the same weights re-read back to back (L2 / Infinity-Cache resident when they fit) vs a rotation
over more weight copies than the caches hold (the decode regime)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
def bench(f, it):
    for _ in range(3): f(0)
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for i in range(it): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for name, N, K, mode in (("o_proj", 3584, 3584, 2), ("qkv", 4608, 3584, 0), ("down", 3584, 18944, 2), ("gate_up", 18944, 3584, 1), ("lm_head", 152064, 3584, 0)):
    rows = 2 * N if mode == 1 else N
    mb = rows * K * 2 / 1e6
    copies = max(2, int(2000 / mb))
    ws = [torch.randn(rows, K, device="cuda").to(torch.bfloat16) * 0.02 for _ in range(copies)]
    x = torch.randn(K, device="cuda").to(torch.bfloat16)
    out = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
    hot = bench(lambda i: ops.gemv(ws[0], x, out, mode), 200)
    cold = bench(lambda i: ops.gemv(ws[i % copies], x, out, mode), 200)
    print(f"{name}: {mb:.1f} MB  same weights {hot:.1f} us ({mb / hot:.2f} TB/s)   rotating {copies} copies {cold:.1f} us ({mb / cold:.2f} TB/s)", flush=True)
    del ws
