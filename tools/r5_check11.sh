#!/bin/bash
# does the vision tower get cheaper per frame when two or three frame groups share ONE pass (larger M for its K = 1280 GEMMs)?
set -u
mkdir -p gpurun_out
{
for nf in 16 32 48 16 32; do QP_VIT_HW=392,560 QP_VIT_FRAMES=$nf python tools/bench_vit.py 2>/dev/null | tail -1; done
for nf in 16 32; do QP_VIT_ARCH=2.5 QP_VIT_HW=392,560 QP_VIT_FRAMES=$nf python tools/bench_vit.py 2>/dev/null | tail -1; done
} | tee gpurun_out/r5k_vit_batching_probe.txt
cd /tmp; export TMPDIR=/tmp
QP_VIT_HW=392,560 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vitprof -o t -- python /root/repo/tools/bench_vit.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
st = sorted(glob.glob("/tmp/vitprof/**/t_kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(st)))
with open("/root/repo/gpurun_out/r5k_vit_cfg4_group_kernel_stats.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- QP_VIT_HW=392,560 python tools/bench_vit.py  (5 passes of one 16-frame cfg4 group through the Qwen2-VL tower)\n")
    f.write("name,calls,total_ns,avg_ns,pct\n")
    for r in rows[:16]:
        n = r["Name"].replace("void ", "").split("(")[0][:100]
        line = f"\"{n}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']}"
        f.write(line + "\n"); print(line)
PY
