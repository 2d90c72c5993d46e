"""Is one [M,3584]x[3584,37888] gate/up GEMM slower than two [.,18944] ones (hipBLASLt picks a different tile for each)?
Sustained loop over the four GEMMs of a layer, like the engine, so that clocks are in the power-limited regime."""
import sys, torch
dev = torch.device("cuda:0")
H, QKV, I = 3584, 4608, 18944
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5760
x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
a = torch.randn(M, I, device=dev, dtype=torch.bfloat16)
wq = torch.randn(QKV, H, device=dev, dtype=torch.bfloat16) * 0.02
wo = torch.randn(H, H, device=dev, dtype=torch.bfloat16) * 0.02
wgu = torch.randn(2 * I, H, device=dev, dtype=torch.bfloat16) * 0.02
wd = torch.randn(H, I, device=dev, dtype=torch.bfloat16) * 0.02
qkv = torch.empty(M, QKV, device=dev, dtype=torch.bfloat16); o = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
gu = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16); g = torch.empty(M, I, device=dev, dtype=torch.bfloat16); u = torch.empty_like(g)
dn = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
def layer(split):
    torch.mm(x, wq.t(), out=qkv); torch.mm(x, wo.t(), out=o)
    if split:
        torch.mm(x, wgu[:I].t(), out=g); torch.mm(x, wgu[I:].t(), out=u)
    else:
        torch.mm(x, wgu.t(), out=gu)
    torch.mm(a, wd.t(), out=dn)
for split in (False, True, False, True):
    for _ in range(10): layer(split)
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(60): layer(split)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 60
    fl = 2 * M * (H * QKV + H * H + 3 * H * I)
    print(f"M={M} gate/up {'2 x N=18944' if split else '1 x N=37888'}: {ms*1e3:.1f} us per layer, {fl/ms/1e9:.0f} TF", flush=True)
