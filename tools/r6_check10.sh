#!/bin/bash
# round 6: qp_patchify (one HIP gather + zero-padded K for the patch embedding) vs the five torch passes, both towers, one group of the 1-hour video
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_e2e_pipeline.py tests/test_gpu_engine.py tests/test_frame_ring.py -m gpu -q -k "patchify or tower or towers or end_to_end or vit or lvu_generate or whichever" 2>&1 | tail -4
{
for rep in 1 2; do
  for hp in 0 1; do
    echo "rep$rep hip_patchify=$hp qwen2-vl   cfg4 group: $(QP_VIT_HIP_PATCHIFY=$hp QP_VIT_HW=392,560 python tools/bench_vit.py 2>/dev/null | tail -1)"
    echo "rep$rep hip_patchify=$hp qwen2.5-vl cfg4 group: $(QP_VIT_ARCH=2.5 QP_VIT_HIP_PATCHIFY=$hp QP_VIT_HW=392,560 python tools/bench_vit.py 2>/dev/null | tail -1)"
    echo "rep$rep hip_patchify=$hp qwen2-vl   cfg2 group: $(QP_VIT_HIP_PATCHIFY=$hp python tools/bench_vit.py 2>/dev/null | tail -1)"
  done
done
} | tee gpurun_out/r6u_hip_patchify_ab.txt
