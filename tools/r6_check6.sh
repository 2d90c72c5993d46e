#!/bin/bash
# round 6: Qwen2.5-VL vision tower: f=0 two GEMMs + silu + mul (torch), f=1 gate | up as ONE zero-padded GEMM + qp_swiglu (torch GEMMs), f=2 the same through the tuned library GEMMs
set -u
mkdir -p gpurun_out
{
for rep in 1 2; do
  for f in 0 1 2; do
    echo "rep$rep mode=$f cfg2 group: $(QP_VIT_ARCH=2.5 QP_VIT25_FUSED_MLP=$((f>0)) QP_VIT25_LT=$((f>1)) python tools/bench_vit.py 2>/dev/null | tail -1)"
    echo "rep$rep mode=$f cfg4 group: $(QP_VIT_ARCH=2.5 QP_VIT25_FUSED_MLP=$((f>0)) QP_VIT25_LT=$((f>1)) QP_VIT_HW=392,560 python tools/bench_vit.py 2>/dev/null | tail -1)"
  done
done
} | tee gpurun_out/r6n_qwen25_tower_tuned_gemms_ab.txt
python -m pytest tests/test_gpu_ops.py tests/test_e2e_pipeline.py -m gpu -q -k "tower or towers or end_to_end or vit" 2>&1 | tail -4
