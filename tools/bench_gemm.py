"""Per-shape GEMM throughput of the decoder's four projections through torch (hipBLASLt), at the group sizes the engine uses."""
import os, sys, torch
dev = torch.device("cuda:0")
H, QKV, I = 3584, 4608, 18944
shapes = [("qkv", H, QKV, True), ("o", H, H, False), ("gate_up", H, 2 * I, False), ("down", I, H, False)]
for M in [int(a) for a in (sys.argv[1:] or ["5760", "2880", "720"])]:
    tot_t = tot_f = 0
    for name, K, N, bias in shapes:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        nt = os.environ.get("QP_GEMM_LAYOUT", "nt") == "nt"               # engine: weights [out, in], used as w.t()
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        w = w.t() if nt else w.t().contiguous()
        b = torch.randn(N, device=dev, dtype=torch.bfloat16) if bias else None
        f = (lambda: torch.addmm(b, x, w)) if bias else (lambda: torch.mm(x, w))
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        fl = 2 * M * K * N
        tot_t += ms; tot_f += fl
        print(f"M={M:5d} {name:8s} K={K:5d} N={N:5d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
    print(f"M={M:5d} layer total {tot_t*1e3:.1f} us  {tot_f/tot_t/1e9:.1f} TF")
