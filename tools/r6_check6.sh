#!/bin/bash
# round 6: Qwen2.5-VL vision tower with gate | up as ONE zero-padded GEMM + qp_swiglu (QP_VIT25_FUSED_MLP=1, default) vs two GEMMs + silu + mul
set -u
mkdir -p gpurun_out
{
for rep in 1 2; do
  for f in 0 1; do
    echo "rep$rep fused_mlp=$f cfg2 group: $(QP_VIT_ARCH=2.5 QP_VIT25_FUSED_MLP=$f python tools/bench_vit.py 2>/dev/null | tail -1)"
    echo "rep$rep fused_mlp=$f cfg4 group: $(QP_VIT_ARCH=2.5 QP_VIT25_FUSED_MLP=$f QP_VIT_HW=392,560 python tools/bench_vit.py 2>/dev/null | tail -1)"
  done
done
} | tee gpurun_out/r6i_qwen25_tower_fused_mlp_ab.txt
python -m pytest tests/test_gpu_ops.py tests/test_e2e_pipeline.py -m gpu -q -k "tower or towers or end_to_end or vit" 2>&1 | tail -4
