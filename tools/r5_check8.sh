#!/bin/bash
# round-5 evidence on one box: the default bench line, rocprofv3 over the same pass, the steady-state window with PMC, the full GPU suite
set -u
mkdir -p gpurun_out
python bench.py > gpurun_out/r5h_bench_line.json 2> gpurun_out/r5h_bench.err; echo "bench rc=$?"; wc -c gpurun_out/r5h_bench_line.json; head -c 600 gpurun_out/r5h_bench_line.json; echo
cp gpurun_out/bench_full.json gpurun_out/r5h_bench_full.json 2>/dev/null
QP_FULLPASS_ARGS="--steps 5 --warmup 2" bash tools/profile_fullpass.sh r5h; echo "fullpass rc=$?"; ls gpurun_out/prof_full_r5h_summary
bash tools/profile_cfg4_window.sh r5h; echo "window rc=$?"; ls gpurun_out/prof_r5h_summary
python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r5h_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -34 gpurun_out/r5h_pytest_gpu.log
