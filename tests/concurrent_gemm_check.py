"""Run by tests/test_gpu_ops.py::test_linear_act_on_two_streams_concurrently in a subprocess (a regression here stalls the device,
which must not take the test session with it).  Two streams issue hipBLASLt GEMMs from one host thread the way pipeline.py does:
the prefill's projections (tuned candidates, M = 2240 and 2255) on one, the vision tower's fc1/fc2 on the other."""
import sys

import torch

from quickvideo_amd.native import QuickPrefillOps

dev = torch.device("cuda:0")
ops = QuickPrefillOps(dev)
g = torch.Generator(device="cuda"); g.manual_seed(11)


def rnd(*shape, s=1.0):
    return (torch.randn(*shape, generator=g, device="cuda") * s).to(torch.bfloat16)


# prefill side: down projection + qkv at two group sizes, 3 "layers" each, tuned like the engine does
pre = []
for m in (2240, 2255):
    for (k, n) in ((18944, 3584), (3584, 4608)):
        x, ws = rnd(m, k), [rnd(n, k, s=0.02) for _ in range(3)]
        out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        ops.linear_tune(x, ws, None, out, ops.ACT_NONE)
        pre.append((x, ws, out, [torch.mm(x, w.t()) for w in ws]))
# vision side: fc1 (Swish epilogue, fp32 bias) and fc2 at the 1-hour video's 8960 patch rows
xv, w1, b1, w2 = rnd(8960, 1280), rnd(5120, 1280, s=0.03), rnd(5120).float(), rnd(1280, 5120, s=0.02)
h_ref = torch.empty(8960, 5120, dtype=torch.bfloat16, device="cuda")
y_ref = torch.empty(8960, 1280, dtype=torch.bfloat16, device="cuda")
ops.linear_act(xv, w1, b1, h_ref, ops.ACT_SWISH, 1.702)
ops.linear_act(h_ref, w2, None, y_ref, ops.ACT_NONE, 1.0 / 1.702)
torch.cuda.synchronize()

sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
hv, yv = torch.empty_like(h_ref), torch.empty_like(y_ref)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    with torch.cuda.stream(sa):
        for x, ws, out, refs in pre:
            for w in ws:
                ops.linear_act(x, w, None, out, ops.ACT_NONE)
    with torch.cuda.stream(sb):
        for _ in range(6):
            ops.linear_act(xv, w1, b1, hv, ops.ACT_SWISH, 1.702)
            ops.linear_act(hv, w2, None, yv, ops.ACT_NONE, 1.0 / 1.702)
    if rep % 10 == 9:
        torch.cuda.synchronize()
        for x, ws, out, refs in pre:                 # `out` holds the last layer's product
            scale = refs[-1].float().abs().max().item()
            bad += int((out.float() - refs[-1].float()).abs().max().item() > 2 ** -7 * scale)
        bad += int(not torch.equal(yv, y_ref)) + int(not torch.equal(hv, h_ref))
torch.cuda.synchronize()
print("concurrent GEMMs: mismatches", bad)
sys.exit(1 if bad else 0)
