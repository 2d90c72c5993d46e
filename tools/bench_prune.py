"""Prune-path micro-benchmark: norm-key form (qp_norm_keys + qp_prune_keys, or qp_prune_keys alone when the RoPE kernel produced
the keys) vs round 1's fused select+gather (qp_prune_staged), back-to-back launches, HIP events."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0")); D = 128

def bench(fn, it=200):
    for _ in range(20): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3

for (n, k, hkv) in [(2240, 1120, 4), (5760, 2880, 4), (2880, 720, 4), (960, 480, 8), (960, 480, 1)]:
    ks = torch.randn(hkv, n, D, device="cuda").bfloat16(); vs = torch.randn_like(ks)
    kc = torch.zeros(hkv, k + 8, D, device="cuda", dtype=torch.bfloat16); vc = torch.zeros_like(kc)
    ss = torch.empty(hkv, n, device="cuda"); ops.key_sumsq(ks, n * D, 0, n, hkv, D, ss)
    idx = torch.empty(k, dtype=torch.int32, device="cuda")
    keys = torch.zeros(n, dtype=torch.int16, device="cuda")
    ops.norm_keys(ss, hkv, n, keys)
    t_old = bench(lambda: ops.prune_staged(ss, hkv, n, k, ks, vs, n * D, hkv, D, kc, vc, (k + 8) * D, 0, idx))
    t_new = bench(lambda: ops.prune_keys(keys, n, k, ks, vs, n * D, hkv, D, kc, vc, (k + 8) * D, 0, idx))
    def two():
        ops.norm_keys(ss, hkv, n, keys); ops.prune_keys(keys, n, k, ks, vs, n * D, hkv, D, kc, vc, (k + 8) * D, 0, idx)
    t_two = bench(two)
    by = n * hkv * D * 2 + 2 * (k * hkv * D * 2 * 2) + 4 * k
    print(f"n={n} k={k} hkv={hkv}: staged {t_old:.1f} us | prune_keys {t_new:.1f} us ({by / t_new / 1e6:.2f} TB/s algorithmic) | norm_keys+prune_keys {t_two:.1f} us")
