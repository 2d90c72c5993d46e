"""Randomised check of the fused decode attention (M-RoPE + KV append + MFMA single-query attention in one launch) against the two
separate entry points (bit-exact) and an fp32 torch reference, over random head configurations and cache lengths (tile / split
boundaries included).  usage: python tools/stress_decode.py [cases]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.native import QuickPrefillOps
ops = QuickPrefillOps(torch.device("cuda:0"))
D = 128
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rnd = random.Random(7)
bad = 0
for ci in range(cases):
    hkv = rnd.choice([1, 2, 4, 8]); G = rnd.choice([1, 2, 3, 4, 5, 6, 7, 8]); hq = hkv * G
    L = rnd.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, rnd.randint(1, 3000), rnd.randint(3000, 40000)])
    cap = L + rnd.randint(0, 5)
    g = torch.Generator(device="cuda"); g.manual_seed(ci)
    kc = torch.randn(hkv, cap, D, generator=g, device="cuda").to(torch.bfloat16); vc = torch.randn(hkv, cap, D, generator=g, device="cuda").to(torch.bfloat16)
    kc2, vc2 = kc.clone(), vc.clone()
    qkv = torch.randn((hq + 2 * hkv) * D, generator=g, device="cuda").to(torch.bfloat16)
    pos = rnd.randint(0, 200000)
    state = torch.tensor([L - 1, pos], dtype=torch.int64, device="cuda")
    cos, sin = ops.mrope_table(torch.full((3, 1), pos, dtype=torch.int64, device="cuda"), (16, 24, 24), 1e6, D)
    ws = ops.decode_attn_workspace(hq, hkv)
    q = torch.empty(hq, D, dtype=torch.bfloat16, device="cuda")
    oa = torch.empty(hq, D, dtype=torch.bfloat16, device="cuda"); ob = torch.empty_like(oa)
    ops.decode_rope_append(qkv, state, 1e6, hq, hkv, D, q, kc, vc, cap * D, cos=cos, sin=sin)
    try:
        ops.decode_attn(q, kc, vc, cap * D, state, hq, hkv, D, D ** -0.5, oa, ws)
    except Exception as e:
        print("case", ci, hq, hkv, L, "decode_attn:", e); bad += 1; continue
    ops.decode_attn_fused(qkv, cos, sin, state, kc2, vc2, cap * D, hq, hkv, D, D ** -0.5, ob, ws)
    torch.cuda.synchronize()
    same = torch.equal(kc.view(torch.int16), kc2.view(torch.int16)) and torch.equal(vc.view(torch.int16), vc2.view(torch.int16)) and \
        torch.equal(oa.view(torch.int16), ob.view(torch.int16))
    kk = kc[:, :L].float().repeat_interleave(G, 0); vv = vc[:, :L].float().repeat_interleave(G, 0)
    sc = torch.einsum("hd,hkd->hk", q.float(), kk) * D ** -0.5
    ref = torch.einsum("hk,hkd->hd", torch.softmax(sc, -1), vv)
    err = (ob.float() - ref).abs()
    ok = bool((err <= 1.5e-2 + 1.5e-2 * ref.abs()).all())
    if not (same and ok):
        bad += 1
        print(f"case {ci}: hq={hq} hkv={hkv} L={L} cap={cap} fused==separate {same} maxerr {err.max().item():.4f}", flush=True)
print(f"{cases} cases, {bad} bad")
