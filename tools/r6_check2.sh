#!/bin/bash
# round 6, GPU session 2: the other BASELINE configurations on the round's code (--lean: the timed pass only), one box; then the HBM-bound
# kernels under rocprofv3 at the cfg2 group shape
mkdir -p gpurun_out
for c in cfg1 cfg2 cfg3 cfg5 cfg4ref cfg4s; do
  python bench.py --config $c --lean --steps 5 --warmup 2 > gpurun_out/r6d_${c}_lean_bench.json 2> gpurun_out/r6d_${c}.log
  python - <<PY
import json
d = json.loads(open("gpurun_out/r6d_${c}_lean_bench.json").read().strip().splitlines()[-1])
print("$c", d["value"], d["unit"], "ms/step", d["ms_per_step"], "attn frac", d.get("roofline", {}).get("frac"))
PY
done
