#!/bin/bash
# Collect PMC counters for bench.py in separate rocprofv3 passes (never combined with trace domains other than kernel-trace).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
P=0
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT GRBM_GUI_ACTIVE"; do
  P=$((P+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $CNT -d "$OUT" -o pmc$P --output-format csv -- python /root/repo/bench.py "$@" > "$OUT/pmc$P.log" 2>&1
done
ls "$OUT"
