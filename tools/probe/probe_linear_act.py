import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
g = torch.Generator(device="cuda"); g.manual_seed(1)
for (m, n, k) in ((23040, 5120, 1280), (77, 64, 32), (5760, 18944, 3584)):
    x = torch.randn(m, k, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(n, generator=g, device="cuda").to(torch.bfloat16)
    y = torch.nn.functional.linear(x.float(), w.float(), b.float())
    for act, alpha, bb, ref in ((0, 1.0, b, y), (1, 1.0, b, torch.nn.functional.silu(y)), (1, 1.702, b.float() * 1.702, 1.702 * y * torch.sigmoid(1.702 * y))):
        out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        ops.linear_act(x, w, bb, out, act, alpha)
        torch.cuda.synchronize()
        err = (out.float() - ref).abs()
        print(m, n, k, "act", act, "max err", err.max().item(), "rel ok", bool((err <= 2.0 ** -7 * ref.abs() + 2e-2).all()))
    # timing vs unfused
    def t(f):
        for _ in range(3): f()
        torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / 10 * 1e3
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    def unf():
        yy = torch.nn.functional.linear(x, w, b); ops.quick_gelu(yy, yy)
    bs = b.float() * 1.702
    print("  fused", t(lambda: ops.linear_act(x, w, bs, out, 1, 1.702)), "us   unfused (linear + quick_gelu)", t(unf), "us")
