#!/bin/bash
# same-box A/B: the flat (stream-K) attention split by cost (default) vs never (QP_ATTN_FLAT=0) on the short-group configurations
set -u
mkdir -p gpurun_out
for rep in 1 2; do for flat in -1 0; do for c in cfg3 cfg2 cfg4ref; do
  QP_ATTN_FLAT=$flat python bench.py --config $c --lean > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab.json"))
print("rep$rep flat=$flat $c", d["value"], d["full_prefill_ms"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
done; done; done | tee gpurun_out/r5j_attn_flat_bench_ab.txt
