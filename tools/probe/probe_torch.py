"""GPU-box environment probe: host cores, hipBLASLt GEMM rates at the 7B shapes, SDPA availability."""
import os, time, torch, json
print("cpu_count", os.cpu_count(), "torch", torch.__version__, "hip", torch.version.hip)
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory / 1e9)
def bench(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(it): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for n in (2240, 2880, 5760):
    for (K, N) in ((3584, 4608), (3584, 3584), (3584, 37888), (18944, 3584)):
        a = torch.randn(n, K, device=dev, dtype=torch.bfloat16); w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        ms = bench(lambda: torch.nn.functional.linear(a, w))
        print(f"gemm n={n} K={K} N={N}: {ms:.3f} ms  {2*n*K*N/ms/1e9:.1f} TF")
# SDPA reference speed (torch's own flash/mem-efficient path), q [1,28,n,128], kv [1,4->28,P+n,128]
for (n, P) in ((5760, 0), (5760, 8640), (2240, 100000)):
    q = torch.randn(1, 28, n, 128, device=dev, dtype=torch.bfloat16)
    k = torch.randn(1, 4, P + n, 128, device=dev, dtype=torch.bfloat16); v = torch.randn_like(k)
    try:
        f = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=(P == 0), enable_gqa=True)
        ms = bench(f, 5)
        fl = 4 * 28 * 128 * (n * P + (n * (n + 1) / 2 if P == 0 else n * n))
        print(f"sdpa n={n} P={P} causal={P==0}: {ms:.3f} ms {fl/ms/1e9:.1f} TF")
    except Exception as ex:
        print("sdpa failed", n, P, repr(ex)[:200])
