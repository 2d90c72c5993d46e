#!/bin/bash
# round 6: rocprofv3 kernel stats of the vision towers alone at the 1-hour video's group shape (8960 patch rows)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for arch in 2 2.5; do
  OUT=/root/repo/gpurun_out/prof_vit_$arch
  QP_VIT_ARCH=$arch QP_VIT_HW=392,560 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python /root/repo/tools/bench_vit.py > $OUT.log 2>&1
  python - <<PY
import csv, glob
st = sorted(glob.glob("$OUT/**/trace_kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(st)))
with open("/root/repo/gpurun_out/r6s_vit_tower_${arch}_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- QP_VIT_ARCH=$arch QP_VIT_HW=392,560 python tools/bench_vit.py  (5 passes of one 16-frame group, 8960 patch rows)\n")
    f.write("name,calls,total_ns,avg_ns,pct\n")
    for r in rows[:25]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
        f.write(f"\"{n}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']}\n")
print(open("/root/repo/gpurun_out/r6s_vit_tower_${arch}_kernel_stats.csv").read())
PY
  rm -rf $OUT
done
