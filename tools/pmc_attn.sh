#!/bin/bash
# usage: tools/pmc_attn.sh <variant> <outdir>   (GPU box) — PMC counters of one attention shape, separate passes
V=${1:-5}; OUT=${2:-/root/repo/gpurun_out/pmc_attn}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export QP_SHAPES=one
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o v${V}_$tag -- python /root/repo/tools/bench_attn.py $V > $OUT/log_$tag.txt 2>&1
done
ls $OUT
