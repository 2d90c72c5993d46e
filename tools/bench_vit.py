"""ViT front-end micro-benchmark (one cfg2 group: 16 frames 560x1008 -> 23040 patches -> 5760 tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickvideo_amd.vit import QWEN2_VL_VIT_7B, QWEN25_VL_VIT_7B, VisionTower, VisionWeights, patchify_frames
dev = torch.device("cuda:0")
w = VisionWeights.synthetic(QWEN25_VL_VIT_7B if os.environ.get('QP_VIT_ARCH') == '2.5' else QWEN2_VL_VIT_7B, dev)
from quickvideo_amd.native import QuickPrefillOps
tower = VisionTower(w, ops=QuickPrefillOps(dev) if os.environ.get('QP_VIT_OPS','1')=='1' else None)
H_, W_ = (int(v) for v in os.environ.get("QP_VIT_HW", "560,1008").split(","))     # QP_VIT_HW=392,560: one group of the 1-hour video (cfg4)
NF = int(os.environ.get("QP_VIT_FRAMES", "16"))                                    # QP_VIT_FRAMES=32: two frame groups in ONE tower pass (M doubles)
frames = torch.randint(0, 256, (NF, 3, H_, W_), dtype=torch.uint8, device=dev)
def f():
    rows, grid = tower.patchify(frames)
    return tower.forward(rows, grid)
for _ in range(2): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(True), torch.cuda.Event(True)
s.record()
for _ in range(3): f()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
npatch, seq = (NF // 2) * (H_ // 14) * (W_ // 14), (H_ // 14) * (W_ // 14)
fl = w.spec.flops_per_patch() * npatch + 32 * (NF // 2) * 16 * 4 * seq * seq * 80
print(f"ViT pass of {NF} frames {H_}x{W_} ({npatch} patch rows): {ms:.2f} ms = {ms * 16 / NF:.2f} ms per 16 frames, {fl/ms/1e9:.1f} TF (linear+attn flops {fl/1e12:.2f} T)")
