#!/bin/bash
# round 6: long soak on the round's library (ten times round 5's counts) + the --full bench legs
set -u
mkdir -p gpurun_out
{
for seed in 31 32 33 34 35 36; do timeout 900 python tools/stress_attn.py 1000 $seed 2>&1 | tail -1; done
timeout 900 python tools/stress_prune_tail.py 30000 2>&1 | tail -1
timeout 600 python tools/stress_decode.py 3000 2>&1 | tail -1
timeout 600 python tools/stress_frame_ring.py 2000 2>&1 | tail -1
} | tee gpurun_out/r6g_long_soak.txt
python bench.py --full > gpurun_out/r6h_cfg4_full_bench_line.json 2> gpurun_out/r6h_stderr.log; cp gpurun_out/bench_full.json gpurun_out/r6h_cfg4_full_bench_full.json; tail -c 600 gpurun_out/r6h_cfg4_full_bench_line.json
