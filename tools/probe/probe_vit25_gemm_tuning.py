"""Qwen2.5-VL vision tower: torch's F.linear (hipBLASLt's first candidate) vs the library's tuned pick (qp_linear_tune) for its four
projections at the two group shapes (8960 rows: a group of the 1-hour video; 23040: a cfg2 group).  Cold weights (4 blocks round-robin)."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quickvideo_amd.native import QuickPrefillOps
dev = torch.device("cuda:0")
ops = QuickPrefillOps(dev)


def bench(f, it=40):
    for i in range(6): f(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for i in range(it): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for M in (8960, 23040):
    for name, K, N in (("qkv", 1280, 3840), ("proj", 1280, 1280), ("gate|up", 1280, 6912), ("down", 3456, 1280)):
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        ws = [(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02) for _ in range(4)]
        b = torch.randn(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_torch = bench(lambda i: F.linear(x, ws[i % 4], b))
        ops.linear_tune(x, ws, b, out)
        choice = ops.linear_plan_choice(M, N, K, 0, b)[0]
        t_lt = bench(lambda i: ops.linear_act(x, ws[i % 4], b, out, ops.ACT_NONE))
        print(f"M={M} {name:8s} K={K} N={N}: F.linear {t_torch:7.1f} us   tuned qp_linear_act {t_lt:7.1f} us (candidate {choice})   {(t_lt / t_torch - 1) * 100:+.1f} %", flush=True)
