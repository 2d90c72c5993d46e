import os, torch, time
dev = torch.device("cuda:0")
def bench(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
shapes = [(5760, 3584, 4608, True), (5760, 3584, 3584, False), (5760, 3584, 37888, False), (5760, 18944, 3584, False)]
res = {}
for mode in ("default", "tunable"):
    if mode == "tunable":
        torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(200); torch.cuda.tunable.set_max_tuning_iterations(30)
    for (n, K, N, bias) in shapes:
        a = torch.randn(n, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, device=dev, dtype=torch.bfloat16); out = torch.empty(n, N, device=dev, dtype=torch.bfloat16)
        f = (lambda: torch.addmm(b, a, w.t(), out=out)) if bias else (lambda: torch.mm(a, w.t(), out=out))
        t0 = time.time(); f(); torch.cuda.synchronize(); t1 = time.time() - t0
        ms = bench(f)
        print(f"{mode} n={n} K={K} N={N}: {ms:.3f} ms {2*n*K*N/ms/1e9:.1f} TF (first call {t1:.1f}s)", flush=True)
