#!/bin/bash
# round 5, first GPU pass: root-cause repro on the round-4 library vs this one, the touched tests, the default bench (compact line)
set -u
mkdir -p gpurun_out
QUICKPREFILL_LIB=$PWD/tools/probe/_parent/libquickprefill_r4.so python tools/probe/repro_ctx_tuner_mismatch.py > gpurun_out/r5_ctx_tuner_r4lib.json 2> gpurun_out/r5_ctx_tuner_r4lib.err
python tools/probe/repro_ctx_tuner_mismatch.py > gpurun_out/r5_ctx_tuner_r5lib.json 2> gpurun_out/r5_ctx_tuner_r5lib.err
python -m pytest tests/test_gpu_engine.py tests/test_gpu_bench.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r5a_pytest1.log 2>&1; echo "pytest1 rc=$?"
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attn or kernel_form or early_out or experiments" > gpurun_out/r5a_pytest2.log 2>&1; echo "pytest2 rc=$?"
fails=0
for i in $(seq 1 12); do
  python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "one_call_segment_path or tuned_gemm_choice" > gpurun_out/r5a_loop_$i.log 2>&1 || fails=$((fails+1))
done
echo "fresh-process loop: $fails failures of 12"
python bench.py > gpurun_out/r5a_bench_line.json 2> gpurun_out/r5a_bench.err; echo "bench rc=$?"
wc -c gpurun_out/r5a_bench_line.json
tail -3 gpurun_out/r5a_pytest1.log gpurun_out/r5a_pytest2.log
