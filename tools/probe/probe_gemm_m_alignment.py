"""Does a tile-aligned ROW count pay for the decoder's four projections?  (round 6 alignment audit.)  The group sizes are ragged in M
(2240 = 8.75 x 256, 5760 = 22.5 x 256, 2880 = 11.25 x 256, 960 = 3.75 x 256); padding the activations to the next multiple of 256 costs
2-7 % more FLOPs.  torch.mm (hipBLASLt's first candidate), cold weights (4 copies), absolute microseconds."""
import torch
dev = torch.device("cuda:0")
H, QKV, I = 3584, 4608, 18944
shapes = [("qkv", H, QKV), ("o", H, H), ("gate_up", H, 2 * I), ("down", I, H)]


def bench(M, K, N, copies=4, it=30):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02).t() for _ in range(copies)]
    for i in range(6): torch.mm(x, ws[i % copies])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for i in range(it): torch.mm(x, ws[i % copies])
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for rep in range(2):
    for m0, m1 in ((2240, 2304), (5760, 5888), (2880, 3072), (960, 1024)):
        tot0 = tot1 = 0.0
        parts = []
        for name, K, N in shapes:
            a, b = bench(m0, K, N), bench(m1, K, N)
            tot0 += a; tot1 += b
            parts.append(f"{name} {a:.0f}/{b:.0f}")
        print(f"rep{rep} M={m0} vs {m1}: layer GEMMs {tot0:.0f} vs {tot1:.0f} us ({(tot1 / tot0 - 1) * 100:+.1f} %)   " + "  ".join(parts), flush=True)
