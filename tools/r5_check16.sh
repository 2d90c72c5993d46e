#!/bin/bash
# does a wider hipBLASLt candidate list (128 instead of 32 heuristic results) hold a faster kernel for the decoder's projections?
set -u
mkdir -p gpurun_out
{
for rep in 1 2; do for tag in product lt128; do
  lib=quickvideo_amd/libquickprefill.so; [ $tag != product ] && lib=tools/experiments/build/libqp_$tag.so
  for c in cfg2 cfg4s; do
    QUICKPREFILL_LIB=$lib QP_LT_DEBUG=1 python bench.py --config $c --lean > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - <<PY
import json,re
d=json.load(open("gpurun_out/ab.json"))
picks=sorted(set(re.findall(r"\[qp_linear_tune\] (m=\d+ n=\d+ k=\d+): candidate (\d+) of (\d+), ([\d.]+) us", open("gpurun_out/ab.err").read())))
print("rep$rep $tag $c", d["value"], d["full_prefill_ms"], "| tuner:", "; ".join(f"{a} -> #{b}/{c_} {t}us" for a,b,c_,t in picks if int(a.split()[0][2:])>=960))
PY
  done; done; done
} | tee gpurun_out/r5s_lt_candidates_ab.txt
