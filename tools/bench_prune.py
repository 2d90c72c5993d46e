"""Prune-path micro-benchmark: fused select+gather (qp_prune_staged) vs separate kernels, back-to-back launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps
D = 128
ops = QuickPrefillOps(torch.device("cuda:0"))
def bench(f, it=200):
    for _ in range(5): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (n, k, hkv) in ((5760, 2880, 4), (2240, 1120, 4), (960, 480, 1)):
    ks = torch.randn(hkv, n, D, device="cuda").to(torch.bfloat16); vs = torch.randn_like(ks)
    ss = torch.empty(hkv, n, dtype=torch.float32, device="cuda")
    ops.key_sumsq(ks, n * D, 0, n, hkv, D, ss)
    kc = torch.zeros(hkv, k + 8, D, dtype=torch.bfloat16, device="cuda"); vc = torch.zeros_like(kc)
    idx = torch.empty(k, dtype=torch.int32, device="cuda")
    t_f = bench(lambda: ops.prune_staged(ss, hkv, n, k, ks, vs, n * D, hkv, D, kc, vc, (k + 8) * D, 0, idx))
    t_s = bench(lambda: ops.select_k_smallest(ss, hkv, n, k, idx))
    t_g = bench(lambda: ops.gather_kv(ks, vs, n * D, idx, k, hkv, D, kc, vc, (k + 8) * D, 0))
    byts = n * hkv * D * 2 + 2 * (k * hkv * D * 2 * 2) + 4 * k
    print(f"n={n} k={k} hkv={hkv}: fused {t_f:.1f} us ({byts/t_f/1e3:.0f} GB/s)  select {t_s:.1f} us  gather {t_g:.1f} us")
