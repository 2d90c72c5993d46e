#!/bin/bash
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attn or kernel_form or early_out or flat" > gpurun_out/r5d_pytest_attn.log 2>&1; echo "pytest attn rc=$?"; tail -4 gpurun_out/r5d_pytest_attn.log
python tools/bench_attn_flat.py > gpurun_out/r5_attn_flat_ab.txt 2>&1; cat gpurun_out/r5_attn_flat_ab.txt
python -m pytest tests/test_gpu_bench.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r5d_pytest_bench.log 2>&1; echo "pytest bench rc=$?"; tail -4 gpurun_out/r5d_pytest_bench.log
