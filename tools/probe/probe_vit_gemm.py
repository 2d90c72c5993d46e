"""Per-GEMM throughput of one ViT block (Qwen2-VL tower: dim 1280, mlp 5120) at the per-group row counts, whole vs row-split."""
import sys, torch
dev = torch.device("cuda:0")
d, m = 1280, 5120
shapes = [("qkv", d, 3 * d), ("proj", d, d), ("fc1", d, m), ("fc2", m, d)]
for M in [int(a) for a in (sys.argv[1:] or ["23040", "11520", "5760"])]:
    tot = 0
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        b = torch.randn(N, device=dev, dtype=torch.bfloat16); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        def run(parts):
            step = M // parts
            for i in range(parts):
                torch.addmm(b, x[i * step:(i + 1) * step], w.t(), out=out[i * step:(i + 1) * step])
        res = []
        for parts in (1, 2, 4):
            for _ in range(3): run(parts)
            torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
            s.record()
            for _ in range(10): run(parts)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / 10)
        fl = 2 * M * K * N
        print(f"M={M} {name:5s} K={K} N={N}: whole {res[0]*1e3:7.1f} us ({fl/res[0]/1e9:6.0f} TF)  2 parts {res[1]*1e3:7.1f}  4 parts {res[2]*1e3:7.1f}", flush=True)
