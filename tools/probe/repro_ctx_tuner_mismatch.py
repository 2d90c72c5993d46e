"""Root cause of the one-call / per-operator bit mismatch (round 4) shown in isolation.

    QUICKPREFILL_LIB=<library> python tools/probe/repro_ctx_tuner_mismatch.py

Context A tunes a GEMM problem (qp_linear_tune: stopwatch over hipBLASLt's heuristic candidates).  Context B is created afterwards on the
same device — what every second QuickPrefillEngine of a process does — and runs qp_linear_act for the same problem WITHOUT tuning it,
exactly as the engine did: its `_lt_tuned` table is per (process, device) and already said "tuned".  With the round-4 library B runs
heuristic candidate 0 whatever A picked, so A's and B's outputs differ in the last bits whenever A's pick was not candidate 0; with the
round-5 library (process-wide choice table) they are equal for every shape.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quickvideo_amd import native  # noqa: E402

import ctypes  # noqa: E402
_lib = ctypes.CDLL(native.LIB_PATH)
for _name in list(native.SIGNATURES):            # an older library (round 4) lacks the entry points added since: bind what it has
    if not hasattr(_lib, _name):
        del native.SIGNATURES[_name]
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
# the tiny engine's projections at the test's segment sizes (398 / 384 / 20 rows) + the 7B down / o projections at cfg4's group size
SHAPES = [(398, 384, 256), (384, 384, 256), (384, 256, 256), (384, 1024, 256), (384, 256, 512), (20, 384, 256), (20, 1024, 256),
          (2240, 3584, 18944), (2240, 3584, 3584), (2240, 4608, 3584), (30, 3584, 18944), (960, 3584, 18944)]
A = native.QuickPrefillOps(dev)
rows = []
for m, n, k in SHAPES:
    x = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ws = [(torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev) for _ in range(3)]
    outA, outB = torch.empty(m, n, dtype=torch.bfloat16, device=dev), torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    A.linear_tune(x, ws, None, outA)
    A.linear_act(x, ws[0], None, outA, A.ACT_NONE)
    B = native.QuickPrefillOps(dev)                      # a later engine's context
    B.linear_act(x, ws[0], None, outB, B.ACT_NONE)       # "already tuned" as far as the host-side table knows
    torch.cuda.synchronize()
    rec = {"m": m, "n": n, "k": k, "bit_equal": bool(torch.equal(outA.view(torch.int16), outB.view(torch.int16))),
           "elements_differ": int((outA.view(torch.int16) != outB.view(torch.int16)).sum())}
    if hasattr(A, "linear_plan_choice") and hasattr(A.lib, "qp_linear_plan_choice"):
        rec["choice_A"], rec["choice_B"] = A.linear_plan_choice(m, n, k)[0], B.linear_plan_choice(m, n, k)[0]
    rows.append(rec)
    del B
print(json.dumps({"library": native.LIB_PATH, "version": A.lib.qp_version().decode(), "all_bit_equal": all(r["bit_equal"] for r in rows), "shapes": rows}))
