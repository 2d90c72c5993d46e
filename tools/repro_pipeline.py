#!/usr/bin/env python3
"""Runs only the video -> first-token leg of bench.py (no timed prefill pass in front) with GPU-side progress on stderr.
usage: QP_PIPELINE_DEBUG=1 python tools/repro_pipeline.py cfg4 [overlapped|sequential]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
mode = sys.argv[2] if len(sys.argv) > 2 else "overlapped"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
spec, cfg, plan, eng, embeds, pos, T = bench.build_workload(name, dev, 0, 1)
del embeds
t0 = time.perf_counter()
res = bench.pipeline_leg(name, eng, dev, modes=(mode,))
print(json.dumps({"wall_s": round(time.perf_counter() - t0, 1), **res}))
