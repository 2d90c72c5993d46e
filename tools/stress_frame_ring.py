"""Stress of the native frame ring behind the plugin (GPU): many videos back to back through ONE pipeline object — ring buffers reused
across videos, producer thread created / joined per video, reader callback and file source alternating, the main stream randomly slowed
so the producer runs into the back-pressure at different points — every run must give the reference tokens.
usage: python tools/stress_frame_ring.py [iterations=120]"""
import os, sys, time, tempfile, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lvu
from quickvideo_amd.engine import QuickPrefillEngine as E
from quickvideo_amd.lvu import load_native_model
from quickvideo_amd.pipeline import PrefillPipeline
from quickvideo_amd.processor import SyntheticProcessor

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
m = load_native_model("synthetic:tiny", device="cuda:0", seed=5)
cfg = lvu.LVUConfig("synthetic:tiny", top_p=0.5, video_group_size=8, num_frames=96)           # 12 groups
d = tempfile.mkdtemp()
frames = np.random.RandomState(7).randint(0, 256, (96, 3, 112, 168), dtype=np.uint8)
np.save(os.path.join(d, "v.npy"), frames)
videos = {"synthetic": "synthetic://?frames=400&h=112&w=168&seed=9", "npy": os.path.join(d, "v.npy")}
pipe = PrefillPipeline(m, cfg, SyntheticProcessor(m.spec))
ref = {k: pipe.generate("What is shown?", v, max_new_tokens=3, overlap=False) for k, v in videos.items()}
orig = E.prefill_group
rng = random.Random(1)
state = {"p": 0.0}

def maybe_slow(self, *a, **kw):
    if rng.random() < state["p"]:
        torch.cuda._sleep(rng.choice((2_000_000, 10_000_000, 30_000_000)))
    return orig(self, *a, **kw)

E.prefill_group = maybe_slow
t0, bad = time.time(), 0
try:
    for i in range(iters):
        kind = ("synthetic", "npy")[i & 1]
        state["p"] = (0.0, 0.3, 1.0)[i % 3]
        os.environ["QP_NATIVE_FILE_SOURCE"] = "0" if (i // 2) & 1 else "1"
        got = pipe.generate("What is shown?", videos[kind], max_new_tokens=3, overlap=(i % 5 != 4))
        if got != ref[kind]:
            bad += 1
            print(f"iteration {i} ({kind}): {got} != {ref[kind]}", flush=True)
finally:
    E.prefill_group = orig
print(f"stress_frame_ring: {iters} videos x 12 groups in {time.time() - t0:.1f} s, mismatches: {bad}")
sys.exit(1 if bad else 0)
