import os, sys, torch
import torch.cuda.tunable as tn
path = sys.argv[1]
tn.enable(True); tn.tuning_enable(True)
tn.set_max_tuning_duration(300); tn.set_max_tuning_iterations(500)
tn.set_filename(path)
dev = torch.device("cuda:0")
H, I = 3584, 18944
for M in (5760, 5775):
    x = torch.randn(M, H, device=dev, dtype=torch.bfloat16)
    w = torch.randn(2 * I, H, device=dev, dtype=torch.bfloat16) * 0.02
    out = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.mm(x, w.t(), out=out)
    torch.cuda.synchronize()
print("done")
