#!/bin/bash
# Quick PMC look at the headline bench (tracked or --exact): lines read, write requests, instructions, wait split.  tools/pmc_quick.sh <out> [bench args]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(realpath -m "$1"); shift; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --ess-batches 0 $@"
for P in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $OUT -o $N --output-format csv -- python $ROOT/bench.py $ARGS > $OUT/$N.log 2>&1
done
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "zz_local" in r["Kernel_Name"] or "zz_general" in r["Kernel_Name"] or "zz_logistic" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(f"{k:28s} n={len(v)} mean_last4={sum(v[-4:])/len(v[-4:]):.5g}")
PY
