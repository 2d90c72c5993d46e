#!/bin/bash
# GPU box: rocprofv3 summaries of a STEADY-STATE WINDOW of the metric's workload (cfg4, the 1-hour video): groups [224, 232) of 450,
# i.e. 2240 new tokens over a ~251k-row pruned prefix per layer, from a fast-forwarded KV arena (bench.py --window).
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, one group per pass (TCC FETCH_SIZE / WRITE_SIZE need separate passes).
# usage: tools/profile_cfg4_window.sh <tag>     -> gpurun_out/prof_<tag>/ ; summaries -> gpurun_out/prof_<tag>/summary/
TAG=${1:-r2}; OUT=/root/repo/gpurun_out/prof_$TAG; WIN=${QP_WINDOW:-224:232}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS="--window $WIN --steps 20 --warmup 1 --lean"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python /root/repo/bench.py $ARGS > $OUT/trace.log 2>&1
if [ -n "$QP_TRACE_ONLY" ]; then python /root/repo/tools/profile_window_summary.py $OUT $TAG "$WIN"; mkdir -p /root/repo/gpurun_out/prof_${TAG}_summary && cp $OUT/summary/* /root/repo/gpurun_out/prof_${TAG}_summary/ && rm -rf $OUT; exit 0; fi
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o pmc_$tag -- python /root/repo/bench.py $ARGS > $OUT/pmc_$tag.log 2>&1
done
python /root/repo/tools/profile_window_summary.py $OUT $TAG "$WIN"
# keep only the summaries (the raw per-dispatch CSVs are hundreds of MB; gpurun merges back at most 64 MiB)
mkdir -p /root/repo/gpurun_out/prof_${TAG}_summary && cp $OUT/summary/* /root/repo/gpurun_out/prof_${TAG}_summary/ && rm -rf $OUT
