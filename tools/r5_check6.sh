#!/bin/bash
# round-5 re-entry baseline: the GPU suite and the default bench line on the restored tree
set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r5f_pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/r5f_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5f_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r5f_smoke.log
python bench.py > gpurun_out/r5f_bench_line.json 2> gpurun_out/r5f_bench.err; echo "bench rc=$?"; wc -c gpurun_out/r5f_bench_line.json; head -c 2500 gpurun_out/r5f_bench_line.json
cp gpurun_out/bench_full.json gpurun_out/r5f_bench_full.json 2>/dev/null
