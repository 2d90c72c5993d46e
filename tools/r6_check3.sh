#!/bin/bash
# end-of-round soak of the kernels on the round's library: randomised attention forms (new seeds), the in-place prune hand-shake,
# the fused decode attention, the frame ring
set -u
mkdir -p gpurun_out
{
for seed in 21 22 23; do timeout 400 python tools/stress_attn.py 300 $seed 2>&1 | tail -1; done
timeout 400 python tools/stress_prune_tail.py 3000 2>&1 | tail -1
timeout 300 python tools/stress_decode.py 500 2>&1 | tail -1
timeout 300 python tools/stress_frame_ring.py 300 2>&1 | tail -1
} | tee gpurun_out/r6e_soak.txt
