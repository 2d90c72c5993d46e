"""Prompt-tail GEMMs (30 rows): x[30,K] @ W[N,K]^T as torch.mm(x, W.t()) vs the transposed problem torch.mm(W, x.t()) -> [N,30]."""
import torch
dev = torch.device("cuda:0")
H, QKV, I = 3584, 4608, 18944
for M in (30, 32, 16):
    for name, K, N in (("qkv", H, QKV), ("o", H, H), ("gate_up", H, 2 * I), ("down", I, H)):
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        o1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); o2 = torch.empty(N, M, device=dev, dtype=torch.bfloat16)
        xt = x.t().contiguous()
        def t(f):
            for _ in range(3): f()
            torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
            s.record()
            for _ in range(20): f()
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / 20 * 1e3
        a = t(lambda: torch.mm(x, w.t(), out=o1)); b = t(lambda: torch.mm(w, x.t(), out=o2)); c = t(lambda: torch.mm(w, xt, out=o2))
        mb = N * K * 2 / 1e6
        print(f"M={M} {name:8s}: x@W^T {a:6.1f} us ({mb/a:4.2f} TB/s)   W@x^T {b:6.1f} us ({mb/b:4.2f} TB/s)   W@xt(contig) {c:6.1f} us ({mb/c:4.2f} TB/s)", flush=True)
