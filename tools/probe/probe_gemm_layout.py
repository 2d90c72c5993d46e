import torch
dev = torch.device("cuda:0")
def bench(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for n in (5760, 2240):
  for (K, N) in ((3584, 4608), (3584, 3584), (3584, 37888), (18944, 3584)):
    a = torch.randn(n, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)          # [N,K] (nn.Linear layout)
    wt = w.t().contiguous()                                          # [K,N]
    out = torch.empty(n, N, device=dev, dtype=torch.bfloat16)
    t1 = bench(lambda: torch.mm(a, w.t(), out=out))
    t2 = bench(lambda: torch.mm(a, wt, out=out))
    outT = torch.empty(N, n, device=dev, dtype=torch.bfloat16)
    t3 = bench(lambda: torch.mm(w, a.t(), out=outT))                  # transposed problem: out^T = W a^T
    fl = 2 * n * K * N / 1e9
    print(f"n={n} K={K} N={N}: W[N,K] {fl/t1:.0f} TF | W[K,N] {fl/t2:.0f} TF | out^T=W a^T {fl/t3:.0f} TF")
