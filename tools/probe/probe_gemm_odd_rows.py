"""Group 0 of a video carries the text prefix: 2240 + 15 = 2255 rows (cfg4), 5760 + 15 = 5775 (cfg2: one group in four).  Do the decoder's
GEMMs pay for a row count that is no multiple of 8 / 16 / 64?  torch.mm (hipBLASLt's first candidate), cold weights, us per layer."""
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
    for ms in ((2240, 2255, 2256, 2304), (5760, 5775, 5776, 5824), (960, 975, 976, 1024)):
        row = []
        for m in ms:
            parts = [bench(m, K, N) for _, K, N in shapes]
            row.append(f"M={m}: {sum(parts):.0f} us (" + "/".join(f"{p:.0f}" for p in parts) + ")")
        print(f"rep{rep}  " + "   ".join(row), flush=True)
