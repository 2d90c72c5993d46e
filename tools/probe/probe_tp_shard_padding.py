"""cfg5's per-rank decoder layer (Qwen2-VL-72B under TP = 8: d = 8192, 8 q + 1 kv head, MLP shard of 29568 / 8 = 3696 columns) through the
engine at the shard width as it is (3696) and as round 6 stores it (zero-padded to 3712 = 29 x 128): seconds per group of n = 960 tokens
over a growing pruned prefix, 4 layers, GEMMs through the engine's own tuned path.  One process, alternating."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quickvideo_amd.engine import QuickPrefillEngine
from quickvideo_amd.lvu_config import LVUConfig
from quickvideo_amd.spec import TextSpec
from quickvideo_amd.weights import DecoderWeights

dev, n, groups, L = "cuda:0", 960, 12, 4
g = torch.Generator(device=dev); g.manual_seed(0)
emb = (torch.randn(n, 8192, generator=g, device=dev) * 0.5).to(torch.bfloat16)
pos = torch.arange(n, device=dev).repeat(3, 1)


def run(inter):
    spec = TextSpec(hidden=8192, n_heads=8, n_kv_heads=1, head_dim=128, intermediate=inter, n_layers=L, vocab=256)
    w = DecoderWeights.synthetic(spec, dev, seed=1)
    eng = QuickPrefillEngine(w, LVUConfig("x", top_p=0.5, video_group_size=16), capacity=groups * n + 64, max_group_tokens=n, device=dev)
    for _ in range(3):                                                  # plan selection / tuning
        eng.prefill_group(emb, pos)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for i in range(groups - 3):
        eng.prefill_group(emb, pos + (i + 3) * n)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (groups - 3) / L * 1e3                  # us per layer and group


for rep in range(3):
    a, b = run(3696), run(3712)
    print(f"rep{rep}: per layer and group (n = 960): I = 3696: {a:.1f} us   I = 3712 (zero-padded): {b:.1f} us   ({(b / a - 1) * 100:+.1f} %)", flush=True)
