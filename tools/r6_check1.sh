#!/bin/bash
# round 6, GPU session 1: the whole GPU suite (no -x), then rocprofv3 over one full pass and over the steady-state window (PMC passes)
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -q) > gpurun_out/r6b_gpu_tests.log 2>&1; tail -6 gpurun_out/r6b_gpu_tests.log
bash tools/profile_fullpass.sh r6 2>&1 | tail -3
bash tools/profile_cfg4_window.sh r6 2>&1 | tail -3
ls gpurun_out/prof_full_r6_summary gpurun_out/prof_r6_summary
