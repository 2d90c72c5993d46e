#!/bin/bash
# HBM read traffic of the decode kernels (GEMV weight stream, single-query attention) from the TCC FETCH_SIZE counter; own pass,
# --kernel-trace only (MI355X_MICROARCH.md HBM section: FETCH_SIZE in KB, x2 correction for wide coalesced streams on gfx950).
OUT=${1:-/root/repo/gpurun_out/pmc_decode}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o gemv -- python /root/repo/tools/bench_gemv.py > $OUT/log_gemv.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o attn -- python /root/repo/tools/bench_decode_attn.py > $OUT/log_attn.txt 2>&1
ls $OUT | head
