import time, torch, os
print("cores", os.cpu_count())
for th in (16, 64, 128):
    torch.set_num_threads(th)
    x = torch.randn(1440, 3584).to(torch.bfloat16); w = torch.randn(18944, 3584).to(torch.bfloat16)
    t=time.perf_counter(); y = torch.nn.functional.linear(x, w); t1=time.perf_counter()-t
    xf, wf = x.float(), w.float()
    t=time.perf_counter(); y = torch.nn.functional.linear(xf, wf); t2=time.perf_counter()-t
    q = torch.randn(1440,128); k = torch.randn(4327,128)
    t=time.perf_counter()
    for h in range(28):
        s = (q @ k.T); s = s.masked_fill(torch.zeros_like(s, dtype=torch.bool), float("-inf")); p = torch.softmax(s, -1); o = p @ k
    t3=time.perf_counter()-t
    t=time.perf_counter(); z = torch.nn.functional.silu(y) * y; t4=time.perf_counter()-t
    print(th, "bf16 linear %.3f fp32 linear %.3f attn28 %.3f silu %.3f" % (t1,t2,t3,t4))
