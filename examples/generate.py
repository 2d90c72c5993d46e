"""The reference's two-line usage (README: `LVU(config).generate(question, video_path)`), on 1 GPU or on N:

    python examples/generate.py                                   # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/generate.py --parallel auto

`--parallel`: tp | sp | pp | auto (a per-video pp x sp grid).  Every rank runs the same script; rank 0 opens the video.
No checkpoint is needed: `synthetic:<preset>` builds seeded random weights at the real dimensions; pass a local Qwen2-VL /
Qwen2.5-VL checkpoint directory as --model to run real weights (and the HF processor as `LVU(cfg, model, processor)`)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from lvu import LVU, LVUConfig  # noqa: E402
from quickvideo_amd.parallel import init_distributed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="synthetic:qwen2-vl-7b")
ap.add_argument("--video", default="synthetic://?frames=512&h=1080&w=1920&fps=2&seed=1")
ap.add_argument("--question", default="Describe what happens in this video.")
ap.add_argument("--num-frames", type=int, default=64)
ap.add_argument("--group-size", type=int, default=16)
ap.add_argument("--top-p", type=float, default=0.5)
ap.add_argument("--max-new-tokens", type=int, default=16)
ap.add_argument("--num-beams", type=int, default=1)
ap.add_argument("--parallel", default=None, choices=[None, "tp", "sp", "pp", "auto", "single"])
args = ap.parse_args()

ctx = init_distributed()                       # no-op in a single-process run
cfg = LVUConfig(model_name_or_path=args.model, model_type="qwen2vl_mi355x", top_k_predict_type="key_norms_small", top_p=args.top_p,
                video_group_size=args.group_size, num_frames=args.num_frames)
lvu = LVU(cfg, model_init_kwargs={"parallel": args.parallel} if args.parallel else {})
out = lvu.generate(args.question, args.video, max_new_tokens=args.max_new_tokens, num_beams=args.num_beams)
if ctx.rank == 0:
    print(out)
