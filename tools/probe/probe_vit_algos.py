"""ViT-tower GEMM shapes at one 16-frame group of the 1-hour video (17 920 patches) and of cfg2 (23 040): torch's default hipBLASLt
pick vs the best heuristic candidate found by qp_linear_tune (rotating weights)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
for M in (17920, 23040):
    for name, K, N, m in (("patch", 1176, 1280, M), ("qkv", 1280, 3840, M), ("proj", 1280, 1280, M), ("fc1", 1280, 5120, M), ("fc2", 5120, 1280, M),
                          ("merge1", 5120, 5120, M // 4), ("merge2", 5120, 3584, M // 4)):
        copies = 8
        ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02 for _ in range(copies)]
        b = torch.randn(N, device="cuda").to(torch.bfloat16) * 0.02
        x = torch.randn(m, K, device="cuda").to(torch.bfloat16); out = torch.empty(m, N, dtype=torch.bfloat16, device="cuda")
        def t(f):
            for i in range(3): f(i)
            torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
            s.record()
            for i in range(24): f(i)
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / 24 * 1e3
        a = t(lambda i: torch.addmm(b, x, ws[i % copies].t(), out=out))
        try:
            ops.linear_tune(x, ws, b, out, 0)
            c = t(lambda i: ops.linear_act(x, ws[i % copies], b, out, 0))
        except Exception as ex:
            c = float("nan")
        fl = 2 * m * K * N
        print(f"M={m} {name:7s} K={K} N={N}: torch.addmm {a:7.1f} us ({fl/a/1e6:5.0f} TF)   tuned candidate {c:7.1f} us ({fl/c/1e6:5.0f} TF)", flush=True)
