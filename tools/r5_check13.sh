#!/bin/bash
# upper-bound probes on the steady-state attention launch (n = 2240 over 255 k rows): product vs builds without the per-element fma /
# row-sum add / both (wrong results by design).  Alternating, two repetitions, same box.
set -u
mkdir -p gpurun_out
B=tools/experiments/build
{
for rep in 1 2; do
  for tag in product nofma nosum nofma_nosum; do
    lib=quickvideo_amd/libquickprefill.so; [ $tag != product ] && lib=$B/libqp_$tag.so
    # scores scaled so that the exponent's argument has the distribution it has in the product (c = scale * log2 e = 0.1275 folded into Q)
    qs=1; [ $tag = nofma ] || [ $tag = nofma_nosum ] && qs=0.1275
    echo "rep$rep $tag: $(QUICKPREFILL_LIB=$lib QP_SHAPES=cfg4 QP_QSCALE=$qs python tools/bench_attn.py 0 2>/dev/null | tail -1)"
  done
done
} | tee gpurun_out/r5m_s6_valu_upper_bounds.txt
