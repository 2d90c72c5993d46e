"""Prompt-tail GEMMs (M = 30) with COLD weights (rotating over 28 layers' worth of copies): torch.mm (hipBLASLt heuristic #0) vs the
i-th heuristic candidate through qp_linear_act (QP_LT_ALGO_INDEX=i per process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
H, QKV, I = 3584, 4608, 18944
M = 30
for name, K, N in (("qkv", H, QKV), ("o", H, H), ("gate_up", H, 2 * I), ("down", I, H)):
    copies = 12
    ws = [torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.02 for _ in range(copies)]
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    def t(f):
        for i in range(3): f(i)
        torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
        s.record()
        for i in range(48): f(i)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 48 * 1e3
    a = t(lambda i: torch.mm(x, ws[i % copies].t(), out=out))
    try:
        b = t(lambda i: ops.linear_act(x, ws[i % copies], None, out, 0))
    except Exception as ex:
        b = float("nan")
    mb = N * K * 2 / 1e6
    print(f"{name:8s} torch.mm {a:6.1f} us ({mb/a:4.2f} TB/s)   candidate {os.environ.get('QP_LT_ALGO_INDEX','0')}: {b:6.1f} us ({mb/b:4.2f} TB/s)", flush=True)
    del ws
