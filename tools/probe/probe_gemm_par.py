import torch
dev = torch.device("cuda:0")
def bench(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
def run(tag, shapes):
    tot_ms = 0; tot_fl = 0
    for (M, K, N) in shapes:
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = bench(lambda: torch.mm(a, w.t(), out=out)); tot_ms += ms; tot_fl += 2 * M * K * N
    print(f"{tag}: layer GEMMs {tot_ms:.3f} ms  -> {tot_fl/tot_ms/1e9:.0f} TF")
n = 5760
for N in (1, 2, 4, 8):
    m = n // N
    run(f"SP N={N} (M={m}, full weights)", [(m, 3584, 4608), (m, 3584, 3584), (m, 3584, 37888), (m, 18944, 3584)])
    hq = 28 // N if 28 % N == 0 else 4
    qkvN = (hq + 2 * max(4 // N, 1)) * 128
    run(f"TP N={N} (M={n}, sharded weights)", [(n, 3584, qkvN), (n, hq * 128, 3584), (n, 3584, 37888 // N), (n, 18944 // N, 3584)])
