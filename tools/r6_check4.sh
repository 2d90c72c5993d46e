#!/bin/bash
# round 6 probe: row sum of the softmax as one v_dot2c_f32_bf16 per packed probability pair (QP_S6_DOT2SUM build) vs the product kernel.
# Same process conditions, alternating, three repetitions; maxerr is against an fp32 reference on the last 512 query rows.
set -u
mkdir -p gpurun_out
B=tools/experiments/build
{
for rep in 1 2 3; do
  for tag in product dot2sum; do
    lib=quickvideo_amd/libquickprefill.so; [ $tag != product ] && lib=$B/libqp_$tag.so
    echo "rep$rep $tag: $(QUICKPREFILL_LIB=$lib QP_SHAPES=cfg4 python tools/bench_attn.py 0 2>/dev/null | tail -1)"
  done
done
for tag in product dot2sum; do
  lib=quickvideo_amd/libquickprefill.so; [ $tag != product ] && lib=$B/libqp_$tag.so
  QUICKPREFILL_LIB=$lib QP_SHAPE="5760,8647,28,4;2240,100000,28,4;960,15000,8,1;2880,2160,28,4" python tools/bench_attn.py 0 2>/dev/null | sed "s/^/$tag: /"
done
} | tee gpurun_out/r6f_s6_dot2sum_ab.txt
