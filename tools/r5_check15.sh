#!/bin/bash
# final check of the round's HEAD: GPU suite, smoke, default bench line
set -u
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r5q_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -3 gpurun_out/r5q_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5q_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5q_smoke.log
python bench.py > gpurun_out/r5q_bench_line.json 2> gpurun_out/r5q_bench.err; echo "bench rc=$?"; wc -c gpurun_out/r5q_bench_line.json; head -c 1900 gpurun_out/r5q_bench_line.json; echo
cp gpurun_out/bench_full.json gpurun_out/r5q_bench_full.json 2>/dev/null
