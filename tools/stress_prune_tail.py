"""Soak test of qp_prune_tail's in-place hand-shake: thousands of launches at several shapes on TWO streams at once while a third
stream keeps every CU busy with GEMMs; every result is compared with the CPU oracle's (kept list + compacted rows).
usage: timeout 300 python tools/stress_prune_tail.py [iterations=2000]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import qp_oracle as O
from quickvideo_amd.native import QuickPrefillOps

D = 128
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ops = QuickPrefillOps(torch.device("cuda:0"))
shapes = [(2887, 5760, 2880, 4), (100, 2240, 1120, 4), (7, 8192, 4096, 2), (0, 960, 480, 8), (33, 8191, 100, 1), (5, 4000, 3999, 4)]
cases = []
for si, (past, n, k, hkv) in enumerate(shapes):
    rs = np.random.RandomState(si)
    keys = torch.from_numpy(rs.standard_normal((hkv, past + n, D)).astype(np.float32)).to(torch.bfloat16)
    vals = torch.from_numpy(rs.standard_normal((hkv, past + n, D)).astype(np.float32)).to(torch.bfloat16)
    ko, vo = O.torch_bf16_to_bits(keys).copy(), O.torch_bf16_to_bits(vals).copy()
    idx, _ = O.prune_tail(ko, vo, past, n, k)
    cases.append(dict(past=past, n=n, k=k, hkv=hkv, keys=keys.cuda(), vals=vals.cuda(), idx=torch.from_numpy(idx.astype(np.int32)).cuda(),
                      wk=torch.from_numpy(ko[:, :past + k].view(np.int16)).cuda(), wv=torch.from_numpy(vo[:, :past + k].view(np.int16)).cuda()))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
load = torch.cuda.Stream()
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
ws = [torch.empty(1 << 20, dtype=torch.uint8, device="cuda") for _ in streams]
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
t0 = time.time()
for it in range(iters):
    if it % 4 == 0:                                   # ~0.9 ms of GEMM per 4 iterations: the prunes always run beside a full machine
        with torch.cuda.stream(load):
            a @ a
    for si, st in enumerate(streams):
        c = cases[(it * 2 + si) % len(cases)]
        with torch.cuda.stream(st):
            kc, vc = c["keys"].clone(), c["vals"].clone()
            idx = torch.full((c["k"],), -1, dtype=torch.int32, device="cuda")
            ops.prune_tail(kc, vc, (c["past"] + c["n"]) * D, c["past"], c["n"], c["k"], c["hkv"], D, idx, ws[si])
            ok = (torch.equal(idx, c["idx"]) and True)
            bad += (~((kc[:, :c["past"] + c["k"]].view(torch.int16) == c["wk"]).all() & (vc[:, :c["past"] + c["k"]].view(torch.int16) == c["wv"]).all()
                      & (idx == c["idx"]).all())).long()
    if it % 200 == 199:
        torch.cuda.synchronize()
        print(f"{it + 1} iterations x 2 streams, {int(bad.item())} mismatches, {time.time() - t0:.1f} s", flush=True)
torch.cuda.synchronize()
print("RESULT", "OK" if int(bad.item()) == 0 else f"{int(bad.item())} MISMATCHES", flush=True)
