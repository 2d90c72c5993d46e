"""Debug: engine vs oracle on odd layer dims, varying one thing at a time (GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import qp_oracle as O
from quickvideo_amd import planner
from quickvideo_amd.engine import QuickPrefillEngine
from quickvideo_amd.lvu_config import LVUConfig
from quickvideo_amd.spec import TextSpec
from quickvideo_amd.weights import DecoderWeights

def run(dims, top_p, groups=2, tune="1", std=0.02):
    os.environ["QP_TUNE_GEMMS"] = tune
    spec, so = TextSpec(**dims), O.TextSpec(**dims)
    w = O.hashed_text_weights(so, seed=21, device="cuda", norm_jitter=0.05, std=std)
    frames, gh, gw, gs, prefix, tail = 8 * groups, 16, 30, 8, 15, 24
    T = prefix + (frames // 2) * (gh // 2) * (gw // 2) + tail
    plan = planner.plan_groups(frames, gs, gh, gw, prefix, T)
    pos, _ = planner.mrope_positions(prefix, (frames // 2, gh, gw), tail)
    embeds = O.hashed_normal((T, spec.hidden), 22, 0.5)
    cfg = LVUConfig("x", top_p=top_p, video_group_size=gs)
    eng = QuickPrefillEngine(DecoderWeights.from_named(spec, w, "cuda:0"), cfg, capacity=T + 8, max_group_tokens=max(plan.tokens + [plan.tail_len]), device="cuda:0")
    eng.kept_trace = []
    post = torch.from_numpy(pos).cuda(); e = embeds.cuda(); st = 0
    for n in plan.tokens:
        eng.prefill_group(e[st:st + n], post[:, st:st + n]); st += n
    lg = eng.prefill_tail(e[st:], post[:, st:]).cpu().numpy()
    ref = O.group_prefill({k: v.cpu() for k, v in w.items()}, so, embeds, pos, plan.tokens, O.PruneCfg(top_p=top_p))
    r = ref["logits"].numpy()
    flat = [k for g in ref["kept"] for k in g]
    diffs = [len(set(a.cpu().numpy().tolist()) ^ set(b.tolist())) // 2 for (l, a), b in zip(eng.kept_trace, flat) if b is not None]
    print("  kept-set differences per (group, layer):", diffs, "std", std)
    print(dims["hidden"], dims["n_heads"], dims["n_kv_heads"], dims["intermediate"], dims["n_layers"], "top_p", top_p, "groups", groups, "tune", tune,
          "len ok", eng.arena.len == ref["cache_len"], "max|d|", float(np.abs(lg - r).max()), "cos", float(np.dot(lg, r) / np.linalg.norm(lg) / np.linalg.norm(r)),
          "|ref|max", float(np.abs(r).max()), flush=True)

base = dict(hidden=8192, n_heads=8, n_kv_heads=1, head_dim=128, intermediate=3696, n_layers=2, vocab=1024)
run(base, 0.5)
run(base, 0.5, std=0.01)
run(dict(base, n_heads=16, n_kv_heads=2), 0.5)
run(dict(base, n_heads=16, n_kv_heads=2), 0.5, std=0.01)
