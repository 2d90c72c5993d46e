"""gate/up GEMM [M,3584]x[3584,37888] (and as two N=18944 GEMMs) over M: which row counts does hipBLASLt serve well?"""
import sys, torch
dev = torch.device("cuda:0")
H, I = 3584, 18944
w = torch.randn(2 * I, H, device=dev, dtype=torch.bfloat16) * 0.02
for M in [int(a) for a in sys.argv[1:]]:
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    gu = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16); g = torch.empty(M, I, device=dev, dtype=torch.bfloat16); u = torch.empty_like(g)
    def t(f):
        for _ in range(3): f()
        torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
        s.record()
        for _ in range(12): f()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 12
    a = t(lambda: torch.mm(x, w.t(), out=gu))
    b = t(lambda: (torch.mm(x, w[:I].t(), out=g), torch.mm(x, w[I:].t(), out=u)))
    fl = 2 * M * H * 2 * I
    print(f"M={M}: fused {a*1e3:7.1f} us ({fl/a/1e9:5.0f} TF)   two {b*1e3:7.1f} us ({fl/b/1e9:5.0f} TF)   us/row {min(a,b)*1e3/M:.4f}", flush=True)
