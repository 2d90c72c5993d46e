"""Does zero-padding a ragged GEMM dimension to a tile multiple pay?  (round 6: the Qwen2.5-VL tower's 3420-wide MLP gained 13 % per group.)
Shapes: the 72B model's per-rank MLP under TP = 8 (I / 8 = 3696 = 28.875 x 128) at cfg5's group size, the vision patch embedding (K = 1176),
and the Qwen2.5 tower's MLP as a cross-check.  torch.mm / addmm (hipBLASLt's first candidate), weights cold (4 copies round-robin)."""
import sys, torch
dev = torch.device("cuda:0")


def bench(M, K, N, bias=False, copies=4, it=40):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    ws = [(torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02).t() for _ in range(copies)]
    b = torch.randn(N, device=dev, dtype=torch.bfloat16) if bias else None
    f = (lambda w: torch.addmm(b, x, w)) if bias else (lambda w: torch.mm(x, w))
    for i in range(8): f(ws[i % copies])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for i in range(it): f(ws[i % copies])
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


cases = [("72B tp8 gate|up  M=960  K=8192", 960, 8192, [2 * 3696, 2 * 3712, 2 * 3840], "N"),
         ("72B tp8 down     M=960  N=8192", 960, 8192, [3696, 3712, 3840], "K"),
         ("72B tp8 gate|up  M=2240 K=8192", 2240, 8192, [2 * 3696, 2 * 3712, 2 * 3840], "N"),
         ("72B tp8 down     M=2240 N=8192", 2240, 8192, [3696, 3712, 3840], "K"),
         ("vit patch embed  M=8960 N=1280", 8960, 1280, [1176, 1216, 1280], "K"),
         ("vit patch embed  M=23040 N=1280", 23040, 1280, [1176, 1216, 1280], "K"),
         ("qwen2.5 vit gate|up M=8960 K=1280", 8960, 1280, [2 * 3420, 2 * 3456], "N"),
         ("qwen2.5 vit down M=8960 N=1280", 8960, 1280, [3420, 3456], "K")]
for rep in range(2):
    for name, M, fixed, vals, which in cases:
        row = []
        for v in vals:
            K, N = (fixed, v) if which == "N" else (v, fixed)
            us = bench(M, K, N)
            row.append(f"{which}={v}: {us:7.1f} us ({2 * M * K * N / us / 1e6:6.1f} TF)")
        print(f"rep{rep} {name}:  " + "   ".join(row), flush=True)
