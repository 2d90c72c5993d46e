#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats over ONE FULL timed pass of the default workload (cfg4, the 1-hour video; --lean = only
# the timed pass: no front-end / CPU / cfg2 legs).  ~140 k dispatches; only the per-kernel stats table is kept.
TAG=${1:-r2}; OUT=/root/repo/gpurun_out/prof_full_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python /root/repo/bench.py --lean ${QP_FULLPASS_ARGS:---steps 20 --warmup 5} > $OUT/bench.json 2> $OUT/trace.log
python - <<PY
import csv, glob, os
st = sorted(glob.glob("$OUT/**/trace_kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(st)))
dst = "/root/repo/gpurun_out/prof_full_${TAG}_summary"; os.makedirs(dst, exist_ok=True)
with open(os.path.join(dst, "${TAG}_cfg4_fullpass_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --lean ${QP_FULLPASS_ARGS:---steps 20 --warmup 5}   (cfg4: one full prefill of the 1-hour video + 5 warm-up steps; 1x MI355X)\n")
    f.write("name,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
    for r in rows[:40]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        f.write(f"\"{n}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
os.system(f"grep '^{{' $OUT/bench.json > {dst}/${TAG}_cfg4_fullpass_bench_under_rocprof.json")
PY
rm -rf $OUT
