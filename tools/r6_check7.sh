#!/bin/bash
# round 6, final check: the whole GPU suite as the driver runs it (-x), smoke(), the default bench
mkdir -p gpurun_out
(time python -m pytest tests -x -q -m gpu) > gpurun_out/r6v_gpu_tests.log 2>&1; tail -5 gpurun_out/r6v_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6v_bench_line.json 2> gpurun_out/r6v_bench_stderr.log; cp gpurun_out/bench_full.json gpurun_out/r6v_bench_full.json; tail -c 1200 gpurun_out/r6v_bench_line.json; wc -c gpurun_out/r6v_bench_line.json
