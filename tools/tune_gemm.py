"""hipBLASLt solution search for the decoder's GEMM shapes through PyTorch TunableOp (library tuning, run on the GPU box).
Result: the search times each candidate in short bursts (boosted clocks, warm MALL: o_proj "1.55 PF"), so its
picks do not hold in the sustained layer loop (+-3 % per shape, net ~0): not adopted.
usage: python tools/tune_gemm.py tune <out.csv> [M ...]     -> writes the TunableOp results file
       python tools/tune_gemm.py eval <file.csv|none> [M ...] -> per-shape TFLOP/s with that file applied (or library defaults)"""
import os, sys, torch
mode, path = sys.argv[1], sys.argv[2]
Ms = [int(a) for a in sys.argv[3:]] or [5760, 5775]
import torch.cuda.tunable as tn
if mode == "tune":
    tn.enable(True); tn.tuning_enable(True)
    tn.set_max_tuning_duration(60); tn.set_max_tuning_iterations(200)
    tn.set_filename(path)
elif path != "none":
    tn.enable(True); tn.tuning_enable(False)
    tn.read_file(path)
dev = torch.device("cuda:0")
H, QKV, I = 3584, 4608, 18944
shapes = [("qkv", H, QKV, True), ("o", H, H, False), ("gate_up", H, 2 * I, False), ("down", I, H, False)]
for M in Ms:
    tot_t = tot_f = 0
    for name, K, N, bias in shapes:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02)
        b = torch.randn(N, device=dev, dtype=torch.bfloat16) if bias else None
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = (lambda: torch.addmm(b, x, w.t(), out=out)) if bias else (lambda: torch.mm(x, w.t(), out=out))
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(30): f()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 30
        fl = 2 * M * K * N
        tot_t += ms; tot_f += fl
        print(f"M={M:5d} {name:8s}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF", flush=True)
    print(f"M={M:5d} layer total {tot_t*1e3:.1f} us  {tot_f/tot_t/1e9:.1f} TF", flush=True)
