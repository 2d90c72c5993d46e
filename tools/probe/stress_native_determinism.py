"""Is the one-call segment path (qp_prefill_segment) deterministic, and bit-equal to the per-operator loop, EVERY time?  The GPU suite's
test_one_call_segment_path_equals_the_per_operator_loop failed once in ~6 runs on the cache rows (kept lists equal).  This runs the
test's comparison many times and reports which pairs ever differ: native vs native (a race inside the call would show here), per-op vs
per-op (run-to-run GEMM nondeterminism would show here) and native vs per-op."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ["QP_GEMM_BACKEND"] = "lt"
os.environ["QP_TUNE_GEMMS"] = "0"
import numpy as np, torch
from test_gpu_engine import make_case, run_gpu, TINY
from quickvideo_amd.lvu_config import LVUConfig

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
spec_o, w, plan, pos, delta, embeds = make_case(24, 16, 24, 8, 15, 20)
KW = (dict(top_p=0.5), dict(top_p=0.5, adaptive_local_attention=False), dict(top_p=0.5, top_k_decay_type="linear", top_k_decay_factor=0.5), dict(top_p=None))


def snap(native, cfg):
    os.environ["QP_NATIVE_SEGMENT"] = native
    eng, logits = run_gpu(TINY, w, plan, pos, embeds, cfg)
    rows = [eng.arena.k(l)[:, :eng.arena.len[l]].cpu().view(torch.int16).numpy().copy() for l in range(3)]
    vrows = [eng.arena.v(l)[:, :eng.arena.len[l]].cpu().view(torch.int16).numpy().copy() for l in range(3)]
    kept = [None if k is None else k.cpu().numpy() for _, k in eng.kept_trace]
    return rows, vrows, kept, logits.numpy().copy()


def diff(a, b):
    out = []
    for l in range(3):
        for nm, x, y in (("k", a[0][l], b[0][l]), ("v", a[1][l], b[1][l])):
            if x.shape != y.shape:
                out.append(f"L{l}{nm}:shape")
            elif not np.array_equal(x, y):
                bad = np.argwhere(x != y)
                toks = np.unique(bad[:, 1])
                out.append(f"L{l}{nm}:{len(bad)} elems, {len(toks)} rows [{toks[:6].tolist()}...], max|d bits|={int(np.max(np.abs(x.astype(np.int32) - y.astype(np.int32))))}")
    if not all((p is None) == (q is None) and (p is None or np.array_equal(p, q)) for p, q in zip(a[2], b[2])):
        out.append("kept")
    if not np.array_equal(a[3], b[3]):
        out.append(f"logits {float(np.max(np.abs(a[3] - b[3]))):.2e}")
    return out


for kw in KW:
    cfg = LVUConfig("x", video_group_size=8, **kw)
    base_n, base_p = snap("1", cfg), snap("0", cfg)
    stats = {"native_vs_native": 0, "perop_vs_perop": 0, "native_vs_perop": 0}
    first = {}
    for r in range(REPS):
        n, p = snap("1", cfg), snap("0", cfg)
        for key, d in (("native_vs_native", diff(n, base_n)), ("perop_vs_perop", diff(p, base_p)), ("native_vs_perop", diff(n, p))):
            if d:
                stats[key] += 1
                first.setdefault(key, d)
    print(kw, stats, first, flush=True)
