#!/bin/bash
# rocprofv3 kernel trace of the decode step (eager + hipGraph legs of tools/bench_decode.py) -> profiles/<tag>_decode_kernel_stats.csv
# usage (on the GPU box): bash tools/profile_decode.sh [tag]      (QP_CFG selects the prefill config, default cfg2)
TAG=${1:-r1}
OUT=/root/repo/gpurun_out/dec
rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o dec -- python /root/repo/tools/bench_decode.py > $OUT/log.txt 2>&1
grep "graph\|eager" $OUT/log.txt
python - "$OUT/dec_results.db" "/root/repo/gpurun_out/${TAG}_decode_kernel_stats.csv" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, grid_x/workgroup_x, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "where name like '%gemv%' or name like '%decode%' or name like '%mrope%' group by name, grid_x order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/bench_decode.py (prefill, then eager and hipGraph greedy decode; 1x MI355X); decode kernels only\n")
    f.write("name,workgroups_x,calls,total_ns,avg_ns,pct_of_decode_kernels,min_ns,max_ns\n")
    for r in rows:
        nm = r[0].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        f.write(f"\"{nm}\",{r[1]},{r[2]},{r[3]},{r[4]:.0f},{r[3] / tot * 100:.1f},{r[5]},{r[6]}\n")
print(open(sys.argv[2]).read())
PY
