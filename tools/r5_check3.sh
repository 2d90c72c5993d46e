#!/bin/bash
set -u
mkdir -p gpurun_out
export QP_BENCH_SINGLE_DEVICE=1
for N in 2 4 8; do
  QP_BENCH_FULL_RECORD=r5_functional_n${N}_one_gpu_cfg2_full.json timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N)) \
    bench.py --gpus $N --steps 2 --warmup 1 --config cfg2 > gpurun_out/r5_functional_n${N}_one_gpu_cfg2.json 2> gpurun_out/r5_functional_n${N}.err
  echo "N=$N rc=$? $(wc -c < gpurun_out/r5_functional_n${N}_one_gpu_cfg2.json) bytes"; grep "\[bench" gpurun_out/r5_functional_n${N}.err | tail -4 | cut -c1-200
done
QP_BENCH_FULL_RECORD=r5_functional_n8_one_gpu_cfg5_full.json timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29520 \
    bench.py --gpus 8 --steps 2 --warmup 1 --config cfg5 --no-pipeline > gpurun_out/r5_functional_n8_one_gpu_cfg5.json 2> gpurun_out/r5_functional_n8_cfg5.err
echo "cfg5 N=8 rc=$?"; grep "\[bench" gpurun_out/r5_functional_n8_cfg5.err | cut -c1-200
QP_SHAPE="960,20000,28,4;2880,20000,28,4;720,60000,28,4;960,216000,28,4" python tools/bench_attn_flat.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_attn_flat_gated.txt
