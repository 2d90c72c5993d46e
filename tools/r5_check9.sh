#!/bin/bash
# the other BASELINE configurations on the round's code (--lean: the timed pass only), + cfg4ref with its front end
set -u
mkdir -p gpurun_out
for c in cfg1 cfg2 cfg3 cfg5 cfg4ref cfg4s; do
  QP_BENCH_FULL_RECORD=r5i_${c}_lean_full.json python bench.py --config $c --lean > gpurun_out/r5i_${c}_lean_bench.json 2> gpurun_out/r5i_${c}_lean.err; echo "$c rc=$?"
  python - <<PY
import json
d=json.load(open("gpurun_out/r5i_${c}_lean_bench.json"))
print("$c", d["value"], d["full_prefill_ms"], d.get("mfma_frac_whole_pass"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("avg_launch_ms"))
PY
done
