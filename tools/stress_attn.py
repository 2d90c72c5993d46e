"""Randomised cross-check of the attention kernel forms on the GPU: s6 (4-/8-wave, planner-chosen splits) against s4 and, on a
row sample, against an fp32 torch reference.  usage: python tools/stress_attn.py [cases] [seed]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps
D = 128
ops = QuickPrefillOps(torch.device("cuda:0"))
cases, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rnd = random.Random(seed)
bad = 0
for ci in range(cases):
    hkv = rnd.choice([1, 2, 4, 8]); hq = hkv * rnd.choice([1, 2, 4, 7])
    n = rnd.choice([1, 7, 33, 64, 127, 128, 129, 255, 256, 300, 511, 720, 1000, 1440, 2240, 2880])
    P = rnd.choice([0, 1, 63, 64, 65, 200, 1000, 4097, 12000, 30000])
    if hq * n * (P + n) > 3.0e9: P = 1000
    sub = rnd.random() < 0.3 and n > 64
    q_row0 = rnd.randrange(0, n - 32) if sub else 0
    nq = rnd.randrange(1, n - q_row0 + 1) if sub else n
    g = torch.Generator(device="cuda"); g.manual_seed(seed * 100003 + ci)
    q = torch.randn(nq, hq, D, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(hkv, P + n, D, generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn(hkv, P + n, D, generator=g, device="cuda").to(torch.bfloat16)
    if rnd.random() < 0.3:                      # spiked keys: force rescales
        k[:, rnd.randrange(P + n)] *= 6
    outs = {}
    for var in ("4", "7", "8", "0"):
        ops.dev_switch("attn_variant", int(var))
        o = torch.full((nq, hq, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.prefill_attn(q, k, v, (P + n) * D, P, k[:, P:], v[:, P:], (P + n) * D, n, hq, hkv, D, D ** -0.5, o, q_row0=q_row0, nq=nq)
        outs[var] = o.float()
    torch.cuda.synchronize()
    # fp32 reference on up to 64 sampled rows
    rows = torch.tensor(sorted(rnd.sample(range(nq), min(nq, 64))), device="cuda")
    kk = k.repeat_interleave(hq // hkv, 0).float(); vv = v.repeat_interleave(hq // hkv, 0).float()
    sc = torch.einsum("rhd,hkd->hrk", q[rows].float(), kk) * D ** -0.5
    lim = (rows + q_row0 + P)[None, :, None]
    sc = sc.masked_fill(torch.arange(P + n, device="cuda")[None, None, :] > lim, float("-inf"))
    ref = torch.einsum("hrk,hkd->rhd", torch.softmax(sc, -1), vv)
    msg = []
    for var in ("4", "7", "8", "0"):
        o = outs[var]
        if not torch.isfinite(o).all():
            msg.append(f"v{var}: non-finite")
            continue
        e = (o[rows] - ref).abs()
        if not (e <= 1.5e-2 + 1.5e-2 * ref.abs()).all():
            msg.append(f"v{var}: err {e.max().item():.4f} vs reference")
        dd = (o - outs["4"]).abs()
        if not (dd <= 4e-3 + 8e-3 * outs["4"].abs()).all():          # one bf16 ulp of the output
            msg.append(f"v{var}: differs from s4 by {dd.max().item():.4f}")
    if msg:
        bad += 1
        print(f"case {ci} n={n} P={P} hq={hq} hkv={hkv} q_row0={q_row0} nq={nq}: " + "; ".join(msg), flush=True)
print(f"{cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
