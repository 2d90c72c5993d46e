#!/bin/bash
# native frame ring on the GPU: its own tests + the pipeline / e2e / two-rank tests that run through it
set -u
mkdir -p gpurun_out
python -m pytest tests/test_frame_ring.py tests/test_e2e_pipeline.py "tests/test_gpu_bench.py::test_bench_two_ranks_on_one_gpu_reproduce_the_single_gpu_token" "tests/test_gpu_engine.py::test_pipeline_ring_reuse_under_a_slow_main_stream" "tests/test_gpu_engine.py::test_lvu_generate_on_gpu" "tests/test_gpu_engine.py::test_generation_kwargs_on_gpu" tests/test_gpu_dist.py -q -m gpu --durations=8 > gpurun_out/r5g_pytest_ring.log 2>&1; echo "ring + pipeline tests rc=$?"; tail -22 gpurun_out/r5g_pytest_ring.log
