"""Group-size GEMMs (M = 5760 / 5775) with rotating (cold) weights: torch.mm vs the best hipBLASLt heuristic candidate found by
qp_linear_tune, then the chosen algorithm re-timed in the same rotating loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["QP_LT_DEBUG"] = "1"
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
H, QKV, I = 3584, 4608, 18944
for M in [int(a) for a in (sys.argv[1:] or ["5760"])]:
    for name, K, N in (("qkv", H, QKV), ("o", H, H), ("gate_up", H, 2 * I), ("gate", H, I), ("down", I, H)):
        copies = 6
        ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02 for _ in range(copies)]
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        def t(f):
            for i in range(3): f(i)
            torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
            s.record()
            for i in range(24): f(i)
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / 24 * 1e3
        a = t(lambda i: torch.mm(x, ws[i % copies].t(), out=out))
        ops.linear_tune(x, ws, None, out, 0)
        b = t(lambda i: ops.linear_act(x, ws[i % copies], None, out, 0))
        fl = 2 * M * K * N
        print(f"M={M} {name:8s} torch.mm {a:7.1f} us ({fl/a/1e6:5.0f} TF)   tuned candidate {b:7.1f} us ({fl/b/1e6:5.0f} TF)", flush=True)
        del ws
