#!/bin/bash
# HBM traffic of the attention kernel from the TCC counters, separate passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
OUT=${1:-/root/repo/gpurun_out/pmc_traffic}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  QP_SHAPES=${QP_SHAPES:-one} timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o t_$tag -- python /root/repo/tools/bench_attn.py 0 > $OUT/log_$tag.txt 2>&1
done
ls $OUT | head
