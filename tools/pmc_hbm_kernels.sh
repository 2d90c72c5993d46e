#!/bin/bash
# GPU box: achieved HBM GB/s of the path's HBM-bound kernels (north_star: key-norm reduction, select, KV gather/compact; plus the
# glue kernels) from rocprofv3: one kernel-trace pass for the durations, one --pmc pass each for FETCH_SIZE and WRITE_SIZE (separate
# passes, MI355X_MICROARCH.md), over `bench.py --config cfg2 --lean` (n = 5760 new tokens per group, k = 2880 kept).
# usage: tools/pmc_hbm_kernels.sh <tag>  -> gpurun_out/hbm_<tag>_summary/<tag>_hbm_kernels.json
TAG=${1:-r2}; CFG=${QP_CFG:-cfg2}; OUT=/root/repo/gpurun_out/hbm_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS=${QP_ARGS:-"--config $CFG --lean --steps 3 --warmup 1"}
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python /root/repo/bench.py $ARGS > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o pmc_$c -- python /root/repo/bench.py $ARGS > $OUT/pmc_$c.log 2>&1
done
python /root/repo/tools/pmc_hbm_summary.py $OUT $TAG $CFG
mkdir -p /root/repo/gpurun_out/hbm_${TAG}_summary && cp $OUT/summary/* /root/repo/gpurun_out/hbm_${TAG}_summary/ && rm -rf $OUT
