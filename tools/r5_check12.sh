#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python tools/stress_frame_ring.py 150 2>&1 | tail -5 | tee gpurun_out/r5l_stress_frame_ring.txt
python -m pytest tests/test_frame_ring.py -q -m gpu 2>&1 | tail -3
