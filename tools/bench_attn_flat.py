"""A/B of the flat (stream-K) split of the prefill-attention launch (round 5): attn_flat = 0 (item-granular plan) / 1 (forced) / -1 (by
cost, the default) on the shapes it is meant for and on the ones it must leave alone.  Same process, alternating, HIP events.
usage: python tools/bench_attn_flat.py  [> gpurun_out/attn_flat_ab.txt]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps

D = 128
ops = QuickPrefillOps(torch.device("cuda:0"))
shapes = [(960, 20000, 28, 4), (960, 108000, 28, 4), (960, 216000, 28, 4), (960, 15000, 8, 1), (960, 4000, 8, 1), (720, 60000, 28, 4),
          (1024, 50000, 28, 4), (2240, 255367, 28, 4), (2240, 20000, 28, 4), (2880, 20000, 28, 4), (5760, 8647, 28, 4), (5760, 0, 28, 4)]
if os.environ.get("QP_SHAPE"):
    shapes = [tuple(int(v) for v in sh.split(",")) for sh in os.environ["QP_SHAPE"].split(";")]


def bench(f, it):
    for _ in range(2): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for (n, P, hq, hkv) in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(n + P)
    q = torch.randn(n, hq, D, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(hkv, P + n, D, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(hkv, P + n, D, generator=g, device="cuda").to(torch.bfloat16)
    out = torch.empty(n, hq, D, dtype=torch.bfloat16, device="cuda")
    fl = 4 * hq * D * (n * P + n * (n + 1) / 2)
    f = lambda: ops.prefill_attn(q, k, v, (P + n) * D, P, k[:, P:], v[:, P:], (P + n) * D, n, hq, hkv, D, D ** -0.5, out)
    it = max(5, min(50, int(2e13 / fl)))
    ops.dev_switch("attn_flat", 0); bench(f, it)                     # warm clocks
    res, outs = {}, {}
    for rep in range(2):
        for mode in (0, 1, -1):
            ops.dev_switch("attn_flat", mode)
            ms = bench(f, it)
            res[mode] = min(res.get(mode, 1e9), ms)
            outs[mode] = out.float().clone()
    d = (outs[0] - outs[1]).abs().max().item()
    tf = {m: fl / res[m] / 1e9 for m in res}
    print(f"n={n} P={P} hq={hq} hkv={hkv}: item-granular {res[0]:.3f} ms ({tf[0]:.0f} TF)  flat {res[1]:.3f} ms ({tf[1]:.0f} TF, {res[0] / res[1]:.3f}x)  "
          f"by-cost {res[-1]:.3f} ms ({tf[-1]:.0f} TF)  max|flat - item-granular| = {d:.4f}", flush=True)
ops.dev_switch("attn_flat", -1)
