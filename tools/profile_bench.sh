#!/bin/bash
# GPU box: rocprofv3 summaries of the default bench command (kernel trace + stats, then PMC traffic passes).
# usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/
TAG=${1:-r1}; OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-pipeline --no-ttft --no-decode"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python /root/repo/bench.py $ARGS > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o pmc_$c -- python /root/repo/bench.py $ARGS > $OUT/pmc_$c.log 2>&1
done
python /root/repo/tools/profile_summary.py $OUT $TAG
