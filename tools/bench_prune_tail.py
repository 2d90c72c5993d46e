"""qp_prune_tail (the public in-place seam) at the cfg2 / cfg4 / cfg5 group shapes: HIP-event time per call, back to back and
with a GEMM between calls (cold instruction cache, like in a layer loop).  Run twice for the A/B of round 3's report:
    python tools/bench_prune_tail.py                          # round 3: norm keys + one in-place launch
    QP_PRUNE_TAIL_STAGED=1 python tools/bench_prune_tail.py   # (library built with `make EXPERIMENTS=1`) round 1/2: sums -> 1-workgroup select -> gather to scratch -> copy back
Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quickvideo_amd.native import QuickPrefillOps  # noqa: E402

D = 128
ops = QuickPrefillOps(torch.device("cuda:0"))
out = {"form": "staged(round1)" if os.environ.get("QP_PRUNE_TAIL_STAGED") else "inplace(round3)", "shapes": {}}
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for name, past, n, k, hkv in (("cfg2", 2887, 5760, 2880, 4), ("cfg4", 250000, 2240, 1120, 4), ("cfg3", 5760, 2880, 720, 4), ("cfg5_rank", 7000, 960, 480, 1)):
    cap = past + n + 8
    kc = torch.randn(hkv, cap, D, device="cuda", dtype=torch.bfloat16)
    vc = torch.randn(hkv, cap, D, device="cuda", dtype=torch.bfloat16)
    idx = torch.empty(k, dtype=torch.int32, device="cuda")
    ws = torch.empty(ops.prune_workspace_bytes(n, k, hkv, D), dtype=torch.uint8, device="cuda")

    def call():
        ops.prune_tail(kc, vc, cap * D, past, n, k, hkv, D, idx, ws)

    for _ in range(5):
        call()
    reps = 200
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        call()
    e.record(); e.synchronize()
    b2b = s.elapsed_time(e) / reps * 1e3
    tot = 0.0
    for _ in range(50):                      # a GEMM in front of every call, the call bracketed alone
        a @ a
        s.record(); call(); e.record(); e.synchronize()
        tot += s.elapsed_time(e) * 1e3
    bytes_alg = n * hkv * D * 2 + 2 * (k * hkv * D * 2 * 2) + 4 * k           # read K_new + read/write kept K,V + indices
    out["shapes"][name] = {"past": past, "n": n, "k": k, "hkv": hkv, "us_back_to_back": round(b2b, 2), "us_after_gemm": round(tot / 50, 2),
                           "algorithmic_MB": round(bytes_alg / 1e6, 2), "GBps_back_to_back": round(bytes_alg / b2b / 1e3, 1)}
print(json.dumps(out))
